/*
 * rdsp_graph.c -- the block-graph runtime of the receive path, in plain C.
 *
 * Mirrors the slice of the Teensy Audio Library "AudioStream" contract that the
 * reference uses: nodes with update(), static connections with
 * fan-out (RadioDSP_SDR_RX.ino:71-89), a ref-counted block pool (AudioMemory(40),
 * .ino:151), receiveReadOnly/release (analyze_fft256iq.cpp:70-71,114-115), one
 * tick = every node's update() in creation order, and the record/play queues
 * that bridge the graph to loop() (RDSP_convolutional.h:205-244,344-349).
 * A block is a tile int16 [n_channels][128]; n_channels = 1 is the reference.
 * Host logic only -- the SDR engine node that launches kernels is in
 * rdsp_graph_sdr.hip.
 */
#include "rdsp_host.h"

#include <stdlib.h>
#include <string.h>

#define RDSP_MAX_PORTS 4
/* ring sizes of the Teensy 4 Audio library as the reference's firmware image has them (AudioRecordQueue::available():
 * `head + 209 - tail`; AudioPlayQueue::playBuffer(): `if (++h > 79) h = 0`): a ring of N holds N - 1 blocks */
#define RDSP_RECORD_QUEUE_MAX 209
#define RDSP_PLAY_QUEUE_MAX 80
#define RDSP_QUEUE_MAX RDSP_RECORD_QUEUE_MAX /* storage */

struct rdsp_block {
  int16_t *data; /* [n_channels][128] */
  int ref_count;
  rdsp_graph_t *graph;
  struct rdsp_block *next_free;
};

typedef struct rdsp_conn {
  rdsp_node_t *dst;
  int src_port, dst_port;
  struct rdsp_conn *next;
} rdsp_conn_t;

struct rdsp_node {
  rdsp_graph_t *graph;
  int ninputs;
  rdsp_block_t *input_queue[RDSP_MAX_PORTS]; /* inputQueueArray */
  rdsp_conn_t *dest_list;                    /* destination_list */
  rdsp_update_fn update;
  void *user;
  void (*destroy_user)(void *);
  struct rdsp_node *next_update; /* creation order */
  /* record / play queue payload (kind != 0) */
  int kind; /* 0 plain, 1 record queue, 2 play queue */
  int enabled;
  rdsp_block_t *fifo[RDSP_QUEUE_MAX];
  int head, tail;
  rdsp_block_t *userblock;
};

struct rdsp_graph {
  int n_channels;
  rdsp_node_t *first_update, *last_update;
  rdsp_block_t *pool;
  int16_t *pool_data;
  int pool_size;
  rdsp_block_t *free_list;
  int blocks_in_use, blocks_in_use_max;
  int irq_disabled;
  unsigned long ticks;
};

rdsp_graph_t *rdsp_graph_create(int n_channels) {
  if (n_channels <= 0) return NULL;
  rdsp_graph_t *g = (rdsp_graph_t *)calloc(1, sizeof(*g));
  if (g) g->n_channels = n_channels;
  return g;
}

void rdsp_graph_destroy(rdsp_graph_t *g) {
  if (!g) return;
  rdsp_node_t *n = g->first_update;
  while (n) {
    rdsp_node_t *nx = n->next_update;
    rdsp_conn_t *c = n->dest_list;
    while (c) {
      rdsp_conn_t *cn = c->next;
      free(c);
      c = cn;
    }
    if (n->destroy_user) n->destroy_user(n->user);
    free(n);
    n = nx;
  }
  free(g->pool);
  free(g->pool_data);
  free(g);
}

int rdsp_graph_channels(const rdsp_graph_t *g) { return g ? g->n_channels : 0; }

/* AudioMemory(n), .ino:151 */
int rdsp_memory(rdsp_graph_t *g, int n_blocks) {
  if (!g || n_blocks <= 0 || g->pool) return RDSP_ERR_INVALID;
  const size_t tile = (size_t)g->n_channels * RDSP_BLOCK_SAMPLES;
  g->pool = (rdsp_block_t *)calloc((size_t)n_blocks, sizeof(rdsp_block_t));
  g->pool_data = (int16_t *)calloc((size_t)n_blocks * tile, sizeof(int16_t));
  if (!g->pool || !g->pool_data) return RDSP_ERR_NOMEM;
  g->pool_size = n_blocks;
  for (int i = n_blocks - 1; i >= 0; i--) {
    g->pool[i].data = g->pool_data + (size_t)i * tile;
    g->pool[i].graph = g;
    g->pool[i].next_free = g->free_list;
    g->free_list = &g->pool[i];
  }
  return RDSP_OK;
}

int rdsp_memory_usage(const rdsp_graph_t *g) { return g ? g->blocks_in_use : 0; }
int rdsp_memory_usage_max(const rdsp_graph_t *g) { return g ? g->blocks_in_use_max : 0; }

/* AudioStream(ninputs, inputQueueArray), analyze_fft256iq.h:55 */
rdsp_node_t *rdsp_node_create(rdsp_graph_t *g, int ninputs, rdsp_update_fn update, void *user) {
  if (!g || ninputs < 0 || ninputs > RDSP_MAX_PORTS) return NULL;
  rdsp_node_t *n = (rdsp_node_t *)calloc(1, sizeof(*n));
  if (!n) return NULL;
  n->graph = g;
  n->ninputs = ninputs;
  n->update = update;
  n->user = user;
  if (g->last_update) g->last_update->next_update = n;
  else g->first_update = n;
  g->last_update = n;
  return n;
}

void rdsp_node_set_destructor(rdsp_node_t *n, void (*fn)(void *)) {
  if (n) n->destroy_user = fn;
}
void *rdsp_node_user(rdsp_node_t *n) { return n ? n->user : NULL; }
rdsp_graph_t *rdsp_node_graph(rdsp_node_t *n) { return n ? n->graph : NULL; }

/* AudioConnection name(src, srcPort, dst, dstPort), .ino:71-89 (fan-out allowed) */
int rdsp_connect(rdsp_node_t *src, int src_port, rdsp_node_t *dst, int dst_port) {
  if (!src || !dst || src->graph != dst->graph || src_port < 0 || src_port >= RDSP_MAX_PORTS ||
      dst_port < 0 || dst_port >= dst->ninputs)
    return RDSP_ERR_INVALID;
  rdsp_conn_t *c = (rdsp_conn_t *)calloc(1, sizeof(*c));
  if (!c) return RDSP_ERR_NOMEM;
  c->dst = dst;
  c->src_port = src_port;
  c->dst_port = dst_port;
  /* append: connections fire in declaration order */
  rdsp_conn_t **pp = &src->dest_list;
  while (*pp) pp = &(*pp)->next;
  *pp = c;
  return RDSP_OK;
}

rdsp_block_t *rdsp_allocate(rdsp_node_t *n) {
  rdsp_graph_t *g = n ? n->graph : NULL;
  if (!g || !g->free_list) return NULL; /* pool exhausted: same NULL as the library */
  rdsp_block_t *b = g->free_list;
  g->free_list = b->next_free;
  b->next_free = NULL;
  b->ref_count = 1;
  g->blocks_in_use++;
  if (g->blocks_in_use > g->blocks_in_use_max) g->blocks_in_use_max = g->blocks_in_use;
  return b;
}

/* release(block), analyze_fft256iq.cpp:114-115 */
void rdsp_release(rdsp_block_t *b) {
  if (!b) return;
  if (b->ref_count > 1) {
    b->ref_count--;
    return;
  }
  rdsp_graph_t *g = b->graph;
  b->ref_count = 0;
  b->next_free = g->free_list;
  g->free_list = b;
  g->blocks_in_use--;
}

int16_t *rdsp_block_data(rdsp_block_t *b) { return b ? b->data : NULL; }
int rdsp_block_refcount(const rdsp_block_t *b) { return b ? b->ref_count : 0; }

/* transmit(block, port): every connection from this port whose destination
 * slot is empty receives a shared reference */
void rdsp_transmit(rdsp_node_t *n, rdsp_block_t *b, int port) {
  if (!n || !b) return;
  for (rdsp_conn_t *c = n->dest_list; c; c = c->next) {
    if (c->src_port != port) continue;
    if (c->dst->input_queue[c->dst_port] == NULL) {
      c->dst->input_queue[c->dst_port] = b;
      b->ref_count++;
    }
  }
}

/* receiveReadOnly(port), analyze_fft256iq.cpp:70-71: ownership of one reference
 * passes to the caller */
rdsp_block_t *rdsp_receive_readonly(rdsp_node_t *n, int port) {
  if (!n || port < 0 || port >= n->ninputs) return NULL;
  rdsp_block_t *b = n->input_queue[port];
  n->input_queue[port] = NULL;
  return b;
}

/* receiveWritable(port): a private copy if the block is shared */
rdsp_block_t *rdsp_receive_writable(rdsp_node_t *n, int port) {
  rdsp_block_t *b = rdsp_receive_readonly(n, port);
  if (b && b->ref_count > 1) {
    rdsp_block_t *p = rdsp_allocate(n);
    if (p)
      memcpy(p->data, b->data, (size_t)n->graph->n_channels * RDSP_BLOCK_SAMPLES * sizeof(int16_t));
    b->ref_count--;
    b = p;
  }
  return b;
}

/* one tick of the audio ISR: every update() in creation order */
int rdsp_update_all(rdsp_graph_t *g) {
  if (!g) return RDSP_ERR_INVALID;
  if (g->irq_disabled) return RDSP_ERR_NOT_READY; /* AudioNoInterrupts() holds the ISR off */
  for (rdsp_node_t *n = g->first_update; n; n = n->next_update)
    if (n->update) n->update(n, n->user);
  g->ticks++;
  return RDSP_OK;
}

/* AudioNoInterrupts() / AudioInterrupts(), .ino:152,175; RDSP_convolutional.h:211,222 */
void rdsp_no_interrupts(rdsp_graph_t *g) { if (g) g->irq_disabled++; }
void rdsp_interrupts(rdsp_graph_t *g) { if (g && g->irq_disabled > 0) g->irq_disabled--; }

/* ---- AudioRecordQueue (graph -> loop) ------------------------------------- */
static void record_update(rdsp_node_t *n, void *user) {
  (void)user;
  rdsp_block_t *b = rdsp_receive_readonly(n, 0);
  if (!b) return;
  if (!n->enabled) {
    rdsp_release(b);
    return;
  }
  int h = (n->head + 1) % RDSP_RECORD_QUEUE_MAX;
  if (h == n->tail) { /* queue full: drop, like the library */
    rdsp_release(b);
    return;
  }
  n->fifo[h] = b;
  n->head = h;
}
rdsp_node_t *rdsp_record_queue_create(rdsp_graph_t *g) {
  rdsp_node_t *n = rdsp_node_create(g, 1, record_update, NULL);
  if (n) n->kind = 1;
  return n;
}
void rdsp_record_queue_begin(rdsp_node_t *q) { /* Q_in_L.begin(), RDSP_convolutional.h:205 */
  if (!q || q->kind != 1) return;
  while (q->tail != q->head) { /* clear() */
    q->tail = (q->tail + 1) % RDSP_RECORD_QUEUE_MAX;
    rdsp_release(q->fifo[q->tail]);
  }
  if (q->userblock) { rdsp_release(q->userblock); q->userblock = NULL; }
  q->enabled = 1;
}
void rdsp_record_queue_end(rdsp_node_t *q) { if (q && q->kind == 1) q->enabled = 0; }
int rdsp_record_queue_available(const rdsp_node_t *q) { /* RDSP_convolutional.h:231 */
  if (!q || q->kind != 1) return 0;
  return (q->head - q->tail + RDSP_RECORD_QUEUE_MAX) % RDSP_RECORD_QUEUE_MAX;
}
int16_t *rdsp_record_queue_readBuffer(rdsp_node_t *q) { /* :236-237 */
  if (!q || q->kind != 1 || q->userblock || q->tail == q->head) return NULL;
  q->tail = (q->tail + 1) % RDSP_RECORD_QUEUE_MAX;
  q->userblock = q->fifo[q->tail];
  return q->userblock->data;
}
void rdsp_record_queue_freeBuffer(rdsp_node_t *q) { /* :243-244 */
  if (!q || q->kind != 1 || !q->userblock) return;
  rdsp_release(q->userblock);
  q->userblock = NULL;
}

/* ---- AudioPlayQueue (loop -> graph) ---------------------------------------- */
static void play_update(rdsp_node_t *n, void *user) {
  (void)user;
  if (n->tail == n->head) return;
  n->tail = (n->tail + 1) % RDSP_PLAY_QUEUE_MAX;
  rdsp_block_t *b = n->fifo[n->tail];
  rdsp_transmit(n, b, 0);
  rdsp_release(b);
}
rdsp_node_t *rdsp_play_queue_create(rdsp_graph_t *g) {
  rdsp_node_t *n = rdsp_node_create(g, 0, play_update, NULL);
  if (n) n->kind = 2;
  return n;
}
int16_t *rdsp_play_queue_getBuffer(rdsp_node_t *q) { /* RDSP_convolutional.h:344-345 */
  if (!q || q->kind != 2) return NULL;
  if (q->userblock) return q->userblock->data;
  q->userblock = rdsp_allocate(q);
  return q->userblock ? q->userblock->data : NULL;
}
int rdsp_play_queue_playBuffer(rdsp_node_t *q) { /* :348-349 */
  if (!q || q->kind != 2 || !q->userblock) return RDSP_ERR_INVALID;
  int h = (q->head + 1) % RDSP_PLAY_QUEUE_MAX;
  if (h == q->tail) return RDSP_ERR_NOT_READY; /* the library spins here; we report */
  q->fifo[h] = q->userblock;
  q->head = h;
  q->userblock = NULL;
  return RDSP_OK;
}

/* ---- input source node (AudioInputI2S role, .ino:52): port 0 = I, 1 = Q ------ */
typedef struct {
  const int16_t *i_tile, *q_tile;
} rdsp_input_state_t;
static void input_update(rdsp_node_t *n, void *user) {
  rdsp_input_state_t *st = (rdsp_input_state_t *)user;
  if (!st->i_tile || !st->q_tile) return; /* nothing captured this tick */
  const size_t bytes = (size_t)n->graph->n_channels * RDSP_BLOCK_SAMPLES * sizeof(int16_t);
  rdsp_block_t *bi = rdsp_allocate(n), *bq = rdsp_allocate(n);
  if (bi && bq) {
    memcpy(bi->data, st->i_tile, bytes);
    memcpy(bq->data, st->q_tile, bytes);
    rdsp_transmit(n, bi, 0);
    rdsp_transmit(n, bq, 1);
  }
  rdsp_release(bi);
  rdsp_release(bq);
  st->i_tile = st->q_tile = NULL;
}
rdsp_node_t *rdsp_input_node_create(rdsp_graph_t *g) {
  rdsp_input_state_t *st = (rdsp_input_state_t *)calloc(1, sizeof(*st));
  if (!st) return NULL;
  rdsp_node_t *n = rdsp_node_create(g, 0, input_update, st);
  if (!n) { free(st); return NULL; }
  n->destroy_user = free;
  return n;
}
/* the tiles [n_channels][128] must stay valid until the next rdsp_update_all() */
int rdsp_input_node_push(rdsp_node_t *n, const int16_t *i_tile, const int16_t *q_tile) {
  if (!n || n->update != input_update) return RDSP_ERR_INVALID;
  rdsp_input_state_t *st = (rdsp_input_state_t *)n->user;
  st->i_tile = i_tile;
  st->q_tile = q_tile;
  return RDSP_OK;
}
