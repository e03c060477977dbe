"""Independent float64 NumPy model of the receive chain (test infrastructure).

Written from the reference's math, not from the C oracle's code, and vectorised
where the oracle loops: used to (a) cross-check the oracle and (b) measure the
float32 noise floor of the recursive stages (NLMS, AGC).  Citations as in
oracle/rdsp_oracle.c.
"""
import numpy as np

BLOCK = 128


def fir_design(n, flo, fhi, fs, window=1):
    """calc_cplx_FIR_coeffs, RDSP_convolutional.h:127-185 (float64)."""
    i = np.arange(n, dtype=np.float64)
    x = i - 0.5 * (n - 1)
    nfl, nfh = flo / fs, fhi / fs
    nfc = (nfh - nfl) / 2.0
    nfs = np.pi * (nfh + nfl)
    t = 2 * np.pi * i / (n - 1)
    if window == 1:
        w = 0.35875 - 0.48829 * np.cos(t) + 0.14128 * np.cos(2 * t) - 0.01168 * np.cos(3 * t)
    elif window == 2:
        w = 0.355768 - 0.487396 * np.cos(t) + 0.144232 * np.cos(2 * t) - 0.012604 * np.cos(3 * t)
    elif window == 3:
        w = np.cos(t / 2)
    elif window == 4:
        w = 0.5 * (1 - np.cos(t))
    else:
        w = 0.3635819 - 0.4891775 * np.cos(t) + 0.1365995 * np.cos(2 * t) - 0.0106411 * np.cos(3 * t)
    with np.errstate(divide="ignore", invalid="ignore"):
        z = np.sin(2 * np.pi * x * nfc) / (np.pi * x) * w
    z[np.abs(x) < 0.01] = 2.0 * nfc
    return z * np.cos(nfs * x), z * np.sin(nfs * x)


def filter_mask(ci, cq, fft_l):
    """init_filter_mask, RDSP_convolutional.h:87-110 (taps narrowed to float32,
    last Q tap cleared by the zero-fill quirk)."""
    h = np.zeros(fft_l, np.complex128)
    nt = fft_l // 2 + 1
    h[:nt] = ci.astype(np.float32).astype(np.float64) + 1j * cq.astype(np.float32).astype(np.float64)
    h[fft_l // 2] = h[fft_l // 2].real
    return np.fft.fft(h)


def lms_mu(strength):
    return 1.0 / (10.0 ** ((strength / 2.0 + 2.0) / 10.0))


class Nlms64:
    """arm_lms_norm_f32 + the delay ring of RDSP_noise_reduction.h:66-80, float64."""

    def __init__(self, strength):
        self.w = np.zeros(96)        # w[i] multiplies state[i], state[0] oldest
        self.init(strength)
        self.calls = 0

    def init(self, strength):
        self.mu = float(np.float32(lms_mu(strength)))
        self.hist = np.zeros(95)
        self.prev = np.zeros(BLOCK)
        self.energy = 0.0
        self.x0 = 0.0

    def block(self, x):
        d = x.copy() if self.calls == 0 else self.prev.copy()
        self.prev = x.copy()
        self.calls += 1
        st = np.concatenate([self.hist, x])
        y = np.empty(BLOCK)
        e = np.empty(BLOCK)
        for n in range(BLOCK):
            win = st[n:n + 96]
            self.energy += x[n] * x[n] - self.x0 * self.x0
            y[n] = win @ self.w
            e[n] = d[n] - y[n]
            g = e[n] * self.mu / (self.energy + 0.000000119209289)
            self.w += g * win
            self.x0 = win[0]
        self.hist = st[BLOCK:]
        return y, e


class Model:
    """One channel, float64.  Same configuration keys as the oracle."""

    def __init__(self, iir_coef=None, **kw):
        """iir_coef: the engine's IIR audio filter as 20 floats ({b0, b1, b2, -a1, -a2} x 4), applied
        to the demodulated audio in float64 (SciPy's sosfilt: the same difference equations)"""
        from oracle_lib import DEFAULTS, DEMOD, AGC, ALS
        self.iir_sos = None
        if iir_coef is not None:
            q = np.asarray(iir_coef, np.float64).reshape(4, 5)
            self.iir_sos = np.stack([q[:, 0], q[:, 1], q[:, 2], np.ones(4), -q[:, 3], -q[:, 4]], axis=1)
            self.iir_zi = np.zeros((4, 2))
        c = dict(DEFAULTS)
        c.update(kw)
        for k, tab in (("demod", DEMOD), ("agc_mode", AGC), ("als_mode", ALS)):
            if isinstance(c[k], str):
                c[k] = tab[c[k]]
        self.c = c
        self.N = c["fft_l"]
        self.H = self.N // 2
        self.D = max(1, c["decim"])
        self.fs_out = c["fs_in"] / self.D
        ci, cq = fir_design(self.H + 1, c["flo_hz"], c["fhi_hz"], self.fs_out, c["window"])
        self.mask = filter_mask(ci, cq, self.N)
        if self.D > 1:
            hi, _ = fir_design(c["fir_taps"], -c["fir_cut_hz"], c["fir_cut_hz"], c["fs_in"], c["window"])
            self.h = hi.astype(np.float32).astype(np.float64)
        self.dphi = int(round(c["nco_hz"] / c["fs_in"] * 2 ** 32)) & 0xFFFFFFFF
        self.n_in = 0
        self.fir_hist = np.zeros(c["fir_taps"] - 1, np.complex128) if self.D > 1 else None
        self.prev = np.zeros(self.H, np.complex128)
        self.pend = np.zeros(0, np.complex128)
        self.nfloor = 0.0
        self.g = 1.0
        self.dc = 0.0
        # SAM (engine feature, build-defined: oracle/rdsp_oracle.c orc_sam_constants / sam_block): loop
        # constants for zeta = 0.65, omegaN = 200 rad/s at the decimated rate, narrowed to float like there
        zeta, wn = 0.65, 200.0
        a = 1.0 - np.exp(-2.0 * wn * zeta / self.fs_out)
        b = -a + 2.0 * (1.0 - np.exp(-wn * zeta / self.fs_out) * np.cos(wn / self.fs_out * np.sqrt(1.0 - zeta * zeta)))
        self.sam_g1, self.sam_g2 = float(np.float32(a)), float(np.float32(b))
        self.sam_wmax = float(np.float32(2.0 * np.pi * 2000.0 / self.fs_out))
        self.sam_phs = self.sam_omega = self.sam_fil = self.sam_dc = 0.0
        self.nr = Nlms64(15)
        self.old_nr = 15
        self.als = Nlms64(c["als_strength"] if c["als_strength"] > 0 else 15)

    def process(self, iq):
        c = self.c
        x = iq[:, 0].astype(np.float64) / 32768.0 * np.float32(c["iq_balance"]) * np.float32(c["input_gain"]) \
            + 1j * (iq[:, 1].astype(np.float64) / 32768.0 * np.float32(c["input_gain"]))
        n = np.arange(self.n_in, self.n_in + len(x), dtype=np.uint64)
        if self.dphi:
            ph = ((n * np.uint64(self.dphi)) & np.uint64(0xFFFFFFFF)).astype(np.float64)
            x = x * np.exp(-2j * np.pi * ph / 2 ** 32)
        if self.D > 1:
            ext = np.concatenate([self.fir_hist, x])
            full = np.convolve(ext, self.h)[len(self.h) - 1:len(self.h) - 1 + len(x)]
            first = (-self.n_in) % self.D
            y = full[first::self.D]
            self.fir_hist = ext[len(ext) - (len(self.h) - 1):]
        else:
            y = x
        self.n_in += len(x)
        self.pend = np.concatenate([self.pend, y])
        outs = []
        while len(self.pend) >= self.H:
            cur, self.pend = self.pend[:self.H], self.pend[self.H:]
            X = np.fft.fft(np.concatenate([self.prev, cur]))
            self.prev = cur
            if c["spectral_nr"] == 2:   # older variant, backup/RadioDSP_SDR_RX_Conv.ino:1586-1630
                mag = np.abs(X)
                lo, hi = 60 * self.N // 256, 120 * self.N // 256
                th = mag[lo:hi + 1].sum() / (hi - lo) * 3.0
                self.nfloor = th
                m1 = np.where(mag <= th, mag * 0.2, mag - th)
                with np.errstate(divide="ignore", invalid="ignore"):
                    X = np.where(mag > 0, X * (m1 / mag), 0.0)
            elif c["spectral_nr"]:
                mag = np.abs(X)
                lo, hi = 30 * self.N // 256, 180 * self.N // 256
                th = mag[lo:hi + 1].sum() / (hi - lo) * (np.float32(c["spectral_level"]) * 1.5)
                self.nfloor = max(self.nfloor + (th - self.nfloor) * float(np.float32(0.65)), 0.0)
                m1 = np.where(mag <= self.nfloor, mag * 0.2, mag - self.nfloor)
                with np.errstate(divide="ignore", invalid="ignore"):
                    X = np.where(mag > 0, X * (m1 / mag), 0.0)
            if c["filter_on"]:
                X = X * self.mask
            yt = np.fft.ifft(X)[self.H:]
            for b in range(0, self.H, BLOCK):
                outs.append(self._post(yt[b:b + BLOCK]))
        return np.concatenate(outs) if outs else np.zeros((0, 2))

    def _post(self, y):
        c = self.c
        L, R = y.real.copy(), y.imag.copy()
        ramp = (np.arange(BLOCK) + 1) / BLOCK
        if c["demod"] == 5:
            a = np.abs(y)
            dn = self.dc + 0.25 * (a.mean() - self.dc)
            L = a - (self.dc + (dn - self.dc) * ramp)
            R = L.copy()
            self.dc = dn
        elif c["demod"] == 6:   # SAM: second-order PLL synchronous detector, serial per sample
            eps = float(np.float32(1e-6))
            for i in range(BLOCK):
                sn, cs = np.sin(self.sam_phs), np.cos(self.sam_phs)
                corr0 = L[i] * cs + R[i] * sn
                corr1 = R[i] * cs - L[i] * sn
                mag2 = corr0 * corr0 + corr1 * corr1
                det = np.arctan2(corr1, corr0) * (mag2 / (mag2 + eps))
                del_out = self.sam_fil
                self.sam_omega = min(max(self.sam_omega + self.sam_g2 * det, -self.sam_wmax), self.sam_wmax)
                self.sam_fil = self.sam_g1 * det + self.sam_omega
                self.sam_phs = (self.sam_phs + del_out) % (2.0 * np.pi)
                self.sam_dc += (corr0 - self.sam_dc) * (1.0 / 512.0)
                L[i] = corr0 - self.sam_dc
            R = L.copy()
        elif c["demod"] != 0:
            R = L.copy()
        if self.iir_sos is not None and c["demod"] != 0:
            from scipy import signal
            L, self.iir_zi = signal.sosfilt(self.iir_sos, L, zi=self.iir_zi)
            R = L.copy()
        if c["lms_nr"] > 0:
            if c["lms_nr"] != self.old_nr:
                self.nr.init(c["lms_nr"])
                self.old_nr = c["lms_nr"]
            yy, _ = self.nr.block(L)
            L = yy * 1.1
            R = L.copy()
        if c["als_mode"]:
            yy, ee = self.als.block(L)
            L = ee if c["als_mode"] == 1 else yy
            R = L.copy()
        if c["agc_mode"]:
            decay = {1: 0.10, 2: 0.03, 3: 0.008}[c["agc_mode"]]
            decay, attack = float(np.float32(decay)), float(np.float32(0.6))
            p = (L * L + R * R).sum() / (2 * BLOCK)
            gt = min(0.25 / (np.sqrt(p) + 1e-6), 100.0)
            gn = self.g + (attack if gt < self.g else decay) * (gt - self.g)
            gi = self.g + (gn - self.g) * ramp
            L, R = L * gi, R * gi
            self.g = gn
        og = 0.0 if c["mute"] else float(np.float32(c["output_gain"]))
        return np.stack([L * og, R * og], axis=1)
