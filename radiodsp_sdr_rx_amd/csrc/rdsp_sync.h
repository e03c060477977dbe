/*
 * rdsp_sync.h -- the workgroup barrier of the kernels, by waves per workgroup (device only).
 */
#ifndef RDSP_SYNC_H
#define RDSP_SYNC_H

#include <hip/hip_runtime.h>

namespace {

/* Workgroup barrier of a kernel whose workgroup is NW waves.  ONE wave needs no instruction at all: its LDS
 * operations execute in order, so a read issued behind a write sees it, and the lanes are synchronised by
 * program order; what remains is to keep the compiler from moving memory accesses across the point, which a
 * wavefront-scope fence does at no cost (LLVM's AMDGPU memory model: no wait for that scope).
 * __syncthreads() in a one-wave workgroup drops the s_barrier but keeps the workgroup-scope fence, an
 * `s_waitcnt lgkmcnt(0)` -- between the writes of a transform pass and the reads of the next one that is a full
 * LDS round trip, 17 times per decimator frame. */
template <int NW>
__device__ __forceinline__ void wg_sync() {
  if constexpr (NW == 1) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}

}  // namespace

#endif
