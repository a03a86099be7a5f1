// Host side of the fused per-snapshot path: launch geometry, the deferred parameter-gradient launch, the slab reduction +
// Adam launch, and the C-ABI entry points (include/gatres.h: gatres_fused_*).
#include <atomic>
#include <vector>
#include "k_fused_dev.h"
#include "k_mask.h"

// the kernels live in translation units of their own (k_window.hip, k_fused_whole.hip); args: const FusedArgs*
extern "C" __attribute__((visibility("hidden"))) int gatres_fused_launch_window(const void* args, int nc, unsigned grid, void* stream);
extern "C" __attribute__((visibility("hidden"))) int gatres_fused_launch_whole(const void* args, int nc, int threads, int cache,
                                                                              unsigned grid, void* stream);

namespace {

static unsigned long long* g_stamps = nullptr;      // gatres_fused_set_stamps (diagnostic build)
static int g_stamp_cap = 0;

// grads = sum of segment slabs (fixed order) ; optionally the Adam update and the loss finalisation in the same pass
__global__ __launch_bounds__(256) void reduce_adam_kernel(const float* __restrict__ slabs, int num_slabs,
                                                          int num_loss, long long stride, long long count,
                                                          float* __restrict__ grads, const float* loss_part,
                                                          float* loss, int do_adam, float* __restrict__ p,
                                                          float* __restrict__ m, float* __restrict__ v,
                                                          unsigned long long* __restrict__ step_counter, double lr,
                                                          double b1, double b2, double eps, double wd,
                                                          float grad_scale, float* __restrict__ wt, int nb, int nc,
                                                          unsigned* __restrict__ status, const double* __restrict__ hp,
                                                          const int* __restrict__ mask_node_ptr, double mask_rate,
                                                          unsigned long long mask_seed, uint8_t* __restrict__ mask_next,
                                                          const unsigned long long* __restrict__ mask_snap, int use_snap,
                                                          int reduce_blocks) {
  __shared__ float s_step_size, s_bc2_sqrt;
  __shared__ unsigned s_fault;
  // The device mask of the NEXT step (utils/auxil.py:166-182), sampled by the workgroups behind the reduction's: the
  // sampler's own launch (5.9 us + a launch boundary per step) disappears from the captured step.  Its key uses the step
  // count this update leaves behind -- exactly what gatres_mask_generate would read at the start of the next step: the
  // count and the fault word as the parameter-gradient launch in front of this one snapshotted them (mask_snap: this
  // launch changes both, and 512 more tickets on the one counter address would cost 10 us).
  if (mask_next && (int)blockIdx.x >= reduce_blocks) {
    const int mb = (int)blockIdx.x - reduce_blocks;
    const unsigned long long next_step = mask_snap[1] == 0ULL ? mask_snap[0] + 1ULL : mask_snap[0];      // (a dropped step: unchanged)
    mask_sample_graph<256, MASK_WGS_UPDATE>(mask_node_ptr, mask_rate, mask_seed, next_step, mask_next, mb / MASK_WGS_UPDATE,
                                            mb % MASK_WGS_UPDATE);
    return;
  }
  if (hp) { lr = hp[0]; b1 = hp[1]; b2 = hp[2]; eps = hp[3]; wd = hp[4]; }      // (gatres_train_step_t.hparams)
  // status[0]: a split launch of this step gave up waiting for a partner workgroup (its results are poisoned).  The step
  // is then DROPPED: no Adam update, no step count, loss = NaN, gradients = NaN; the last block clears the word and
  // counts the event in status[1], so one transient stall costs one step instead of the whole run.
  // snap: the parameter-gradient launch in front of this one left the step count and the fault word in mask_snap (the training
  // step's sequence, gatres_train_step).  Every block then reads THOSE -- stable words -- and block 0 alone counts the step:
  // no tickets (258 read-modify-writes on one address serialise in the L2 at ~20 ns each: 5 of this launch's 8 us).
  const bool snap = use_snap != 0;
  if (threadIdx.x == 0)
    s_fault = snap ? (unsigned)mask_snap[1] : (status ? __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u);
  if (do_adam && threadIdx.x == 0) {
    const unsigned long long t =
        (snap ? mask_snap[0] : __hip_atomic_load(step_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) + 1ULL;
    s_step_size = (float)(lr / (1.0 - gatres_powi(b1, t)));
    s_bc2_sqrt = (float)sqrt(1.0 - gatres_powi(b2, t));
    if (snap && s_fault == 0u && blockIdx.x == 0) step_counter[0] = t;      // (nobody reads the counter in this launch)
    // The step is counted once every block has READ the counter: a ticket drawn right after this block's read (its value is
    // in a register: the wait below), the last ticket increments.  Nothing orders the count behind the parameter stores --
    // the next launch is -- so no fence: the ticket used to follow the block's stores behind a __threadfence(), an L2
    // write-back + invalidate of ~3.5 us at the end of every block of a 14-us launch.
    if (!snap) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (!snap && s_fault == 0u) {
      const unsigned long long done = atomicAdd(&step_counter[1], 1ULL);
      if (done == (unsigned long long)reduce_blocks - 1ULL) {
        step_counter[1] = 0ULL;
        atomicAdd(&step_counter[0], 1ULL);
      }
    }
  }
  if (loss_part && blockIdx.x == 0 && threadIdx.x < 64) {
    // the loss: sum of the per-(segment, part) squared errors / masked-node count.  One wave, strided partial sums
    // and a fixed-order butterfly (a single thread walking 128 dependent loads made block 0 the kernel's critical path)
    float s = 0.f;
    for (int k = threadIdx.x; k < num_loss; k += 64) s += loss_part[k];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (threadIdx.x == 0) loss[0] = s / loss_part[num_loss];
  }
  __syncthreads();
  const bool fault = s_fault != 0u;
  if (fault && loss && blockIdx.x == 0 && threadIdx.x == 0) loss[0] = NAN;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx < count && (int)blockIdx.x < reduce_blocks) {
    float acc = 0.f;
    int s0 = 0;
    // the parameter and its moments are requested together with the first slab rows: one memory round trip for a
    // 32-snapshot batch instead of five dependent ones (this launch is pure latency: 258 workgroups x 4 waves)
    float pv = 0.f, mv0 = 0.f, vv0 = 0.f;
    if (do_adam && !fault) { pv = p[idx]; mv0 = m[idx]; vv0 = v[idx]; }
    for (; s0 + 32 <= num_slabs; s0 += 32) {            // 32 loads in flight, summed in slab order
      float v32[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) v32[u] = slabs[(size_t)(s0 + u) * stride + idx];
#pragma unroll
      for (int u = 0; u < 32; ++u) acc += v32[u];
    }
    for (; s0 + 8 <= num_slabs; s0 += 8) {
      float v8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v8[u] = slabs[(size_t)(s0 + u) * stride + idx];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v8[u];
    }
    for (; s0 < num_slabs; ++s0) acc += slabs[(size_t)s0 * stride + idx];
    grads[idx] = fault ? NAN : acc;
    if (do_adam && !fault) {
      float gv = acc * grad_scale;
      gv = gv + (float)wd * pv;
      float mv = mv0;
      mv = mv + (float)(1.0 - b1) * (gv - mv);
      const float vv = (float)b2 * vv0 + (float)(1.0 - b2) * gv * gv;
      const float denom = sqrtf(vv) / s_bc2_sqrt + (float)eps;
      const float pn = pv + (-s_step_size * mv) / denom;
      p[idx] = pn;
      m[idx] = mv;
      v[idx] = vv;
      if (wt) {                   // keep the transposed conv weights of the next backward current (k_misc.hip layout)
        const long long per = 2LL * nc * nc, stride = 9LL * nc + 2 * per, off = idx - 2LL * nc;
        if (off >= 0 && off < (long long)nb * stride) {
          const long long b = off / stride, o = off % stride;
          if (o >= 6LL * nc && o < 6LL * nc + per) {                        // W1 [2nc][nc] -> [nc][2nc]
            const long long e = o - 6LL * nc, row = e / nc, col = e % nc;
            wt[b * 2 * per + col * 2 * nc + row] = pn;
          } else if (o >= 9LL * nc + per) {                                 // W2 [nc][2nc] -> [2nc][nc]
            const long long e = o - 9LL * nc - per, row = e / (2 * nc), col = e % (2 * nc);
            wt[b * 2 * per + per + col * nc + row] = pn;
          }
        }
      }
    }
  }
  if (fault && snap && blockIdx.x == 0 && threadIdx.x == 0) {      // (the other blocks read the snapshot, not the word)
    status[1] += 1u;
    __hip_atomic_store(status, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (fault && !snap && threadIdx.x == 0) {      // every block has read status[0] before the last ticket is drawn
    __threadfence();
    if (atomicAdd(status + 2, 1u) == (unsigned)reduce_blocks - 1u) {
      status[2] = 0u;
      status[1] += 1u;
      __hip_atomic_store(status, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Parameter gradients of a RANGE of blocks, summed over the segments, in ONE launch: what the data-parallel step of the
// fused path forms its gradient buckets with (gatres_train_step: GATRES_FLAG_GRADS_ONLY) -- the upper blocks' bucket goes on
// the wire while the lower blocks' launch runs.
// A COLUMN is one (block, conv): the S = num_segments items of a column (one workgroup each, param_grads_stream_kernel's
// items) write the column's range of S slab rows with agent-scope stores (sc1: written through), acknowledged (every wave's
// s_waitcnt vmcnt(0)) before the item counts itself at the column's counter; the item that ARRIVES LAST reads the rows
// back with agent-scope loads (never served from its L1 or a stale L2 line: the granule protocol's pair of accesses,
// k_fused_dev.h), sums them in slab order -- reduce_adam_kernel's association: the same bits, whoever arrives last --,
// writes grads and, if asked (do_adam: the whole model only), applies Adam to those parameters.  lin0's range rides on
// column (block 0, conv1), lin1's on (last block, conv2).
// As a replacement of the single-GPU step's two launches (param_grads_stream_kernel 37.6 us + reduce_adam_kernel 8.7 us
// inside the captured step) it was measured and is NOT used: 53 us with everything in it (47.7 without Adam) -- the column
// sums are a serial tail of dependent memory round trips on the workgroups that finish last; a release fence per arriver
// instead of agent-scope stores 73 us (960 L2 write-back walks); dedicated reducer workgroups behind the items 60 us
// (a third round of workgroups): profiles/r04_param_grads_finish_probe.txt, tests/micro/pgf_probe.py.
// A faulted split launch (status[0]) turns grads and loss into NaN and skips the update, as gatres_fused_finish;
// `final_launch`: this launch holds block 0 (it clears the fault).
struct FinishArgs {
  float* grads;
  const float* loss_part;
  float* loss;
  int num_loss, do_adam;
  float *p, *m, *v;
  unsigned long long* step_counter;
  double lr, b1, b2, eps, wd;
  const double* hp;
  float grad_scale;
  float* wt;
  unsigned* status;
  unsigned* colcnt;
  int S, b_lo, b_hi, final_launch;
};

template <int NC>
__global__ __launch_bounds__(PGR_THREADS) void param_grads_finish_kernel(const ParamGradArgs a, const FinishArgs f) {
  __shared__ __attribute__((aligned(16))) float lds[PGR_LDS_FLOATS(NC)];
  __shared__ int s_last;
  __shared__ float s_step_size, s_bc2_sqrt;
  __shared__ unsigned s_fault;
  const Layout& L = a.L;
  const int S = f.S, tid = threadIdx.x;
  // segment-major: a segment's items are adjacent workgroups, every column completes with the launch's last segment
  // (column-major measured 56 us against 53)
  const int ncol = 2 * (f.b_hi - f.b_lo);
  const int col = (int)(blockIdx.x % ncol), seg = (int)(blockIdx.x / ncol);
  const int b = f.b_hi - 1 - (col >> 1), conv = 1 - (col & 1);          // production order: upper blocks first, conv2 before conv1
  if (f.loss_part && blockIdx.x == 0 && tid < 64) {                      // the loss (as reduce_adam_kernel)
    float s = 0.f;
    for (int k = tid; k < f.num_loss; k += 64) s += f.loss_part[k];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const bool flt = f.status && __hip_atomic_load(f.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    if (tid == 0) f.loss[0] = flt ? NAN : s / f.loss_part[f.num_loss];
  }
  if (conv == 0) param_grads_item_reg<NC, 0, true>(a, seg, b, lds);
  else           param_grads_item_reg<NC, 1, true>(a, seg, b, lds);
  const bool has_lin0 = b == 0 && conv == 0, has_lin1 = b == L.nb - 1 && conv == 1;
  if (a.M > 1) {                                                         // lin0 / lin1 partials of a split segment
    if (has_lin0) fold_parts<PGR_THREADS, true>(a, seg, L.p_lin0_w, 2 * NC);
    if (has_lin1) fold_parts<PGR_THREADS, true>(a, seg, L.p_lin1_w, NC + 1);
  }
  // ---- arrive at the column's counter (no fence: see above)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned* cnt = f.colcnt + (2 * b + conv);
  if (tid == 0) {
    const unsigned prev = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = prev == (unsigned)(S - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  double lr = f.lr, b1 = f.b1, b2 = f.b2, eps = f.eps, wd = f.wd;
  if (f.hp) { lr = f.hp[0]; b1 = f.hp[1]; b2 = f.hp[2]; eps = f.hp[3]; wd = f.hp[4]; }
  if (tid == 0) {
    __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);           // (the next launch starts from zero)
    s_fault = f.status ? __hip_atomic_load(f.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    if (f.do_adam) {
      const unsigned long long t = __hip_atomic_load(f.step_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ULL;
      s_step_size = (float)(lr / (1.0 - gatres_powi(b1, t)));
      s_bc2_sqrt = (float)sqrt(1.0 - gatres_powi(b2, t));
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // (the step count and the fault word are in registers)
  }
  __syncthreads();
  const bool fault = s_fault != 0u;
  const int64_t po = L.p_block0 + (int64_t)b * L.p_block_stride;
  const __amdgpu_buffer_rsrc_t slab_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(a.slabs, 0, (int)((size_t)S * L.slab_stride * 4), 0x00020000);
  typedef float v4f __attribute__((ext_vector_type(4)));
  auto slab4 = [&](int srow, int64_t idx) -> float4 {                   // 16 bytes of slab row srow, agent scope (sc1)
    const v4f v = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(
        slab_rsrc, (unsigned)(((size_t)srow * L.slab_stride + idx) * 4), 0, 16));
    return make_float4(v.x, v.y, v.z, v.w);
  };
  // one parameter: Adam on the summed gradient (the arithmetic of reduce_adam_kernel, statement for statement)
  auto update = [&](int64_t idx, float acc, float pv, float mv0, float vv0) {
    f.grads[idx] = fault ? NAN : acc;
    if (f.do_adam && !fault) {
      float gv = acc * f.grad_scale;
      gv = gv + (float)wd * pv;
      float mv = mv0;
      mv = mv + (float)(1.0 - b1) * (gv - mv);
      const float vv = (float)b2 * vv0 + (float)(1.0 - b2) * gv * gv;
      const float denom = sqrtf(vv) / s_bc2_sqrt + (float)eps;
      const float pn = pv + (-s_step_size * mv) / denom;
      f.p[idx] = pn;
      f.m[idx] = mv;
      f.v[idx] = vv;
      if (f.wt) {               // keep the transposed conv weights of the next backward current (k_misc.hip layout)
        const long long per = 2LL * NC * NC, o = idx - po;
        if (o >= 6LL * NC && o < 6LL * NC + per) {                          // W1 [2nc][nc] -> [nc][2nc]
          const long long e = o - 6LL * NC, row = e / NC, cl = e % NC;
          f.wt[(long long)b * 2 * per + cl * 2 * NC + row] = pn;
        } else if (o >= 9LL * NC + per && o < 9LL * NC + 2 * per) {         // W2 [nc][2nc] -> [2nc][nc]
          const long long e = o - 9LL * NC - per, row = e / (2 * NC), cl = e % (2 * NC);
          f.wt[(long long)b * 2 * per + per + cl * NC + row] = pn;
        }
      }
    }
  };
  // [lo, hi) of the flat vector, lo % 4 == 0: 16 bytes per thread and slab row, twelve rows in flight (sixteen spill at the
  // 128 VGPRs that two workgroups per CU leave), every element summed in slab order
  auto finish_range = [&](int64_t lo, int64_t hi) {
    const int64_t n4 = (hi - lo) >> 2;
    for (int64_t q = tid; q < n4; q += PGR_THREADS) {
      const int64_t idx = lo + 4 * q;
      float4 pv = f4zero(), mv0 = f4zero(), vv0 = f4zero();
      if (f.do_adam && !fault) { pv = ld4(f.p + idx); mv0 = ld4(f.m + idx); vv0 = ld4(f.v + idx); }
      float4 acc = f4zero();
      int s0 = 0;
      for (; s0 + 12 <= S; s0 += 12) {
        float4 v12[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) v12[u] = slab4(s0 + u, idx);
#pragma unroll
        for (int u = 0; u < 12; ++u) { acc.x += v12[u].x; acc.y += v12[u].y; acc.z += v12[u].z; acc.w += v12[u].w; }
      }
      for (; s0 + 8 <= S; s0 += 8) {
        float4 v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v8[u] = slab4(s0 + u, idx);
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += v8[u].x; acc.y += v8[u].y; acc.z += v8[u].z; acc.w += v8[u].w; }
      }
      for (; s0 < S; ++s0) {
        const float4 v1 = slab4(s0, idx);
        acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
      }
      update(idx, acc.x, pv.x, mv0.x, vv0.x);
      update(idx + 1, acc.y, pv.y, mv0.y, vv0.y);
      update(idx + 2, acc.z, pv.z, mv0.z, vv0.z);
      update(idx + 3, acc.w, pv.w, mv0.w, vv0.w);
    }
    for (int64_t idx = lo + 4 * n4 + tid; idx < hi; idx += PGR_THREADS) {      // (lin1's odd element)
      float pv = 0.f, mv0 = 0.f, vv0 = 0.f;
      if (f.do_adam && !fault) { pv = f.p[idx]; mv0 = f.m[idx]; vv0 = f.v[idx]; }
      float acc = 0.f;
      for (int s0 = 0; s0 < S; ++s0)
        acc += __hip_atomic_load(a.slabs + (size_t)s0 * L.slab_stride + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      update(idx, acc, pv, mv0, vv0);
    }
  };
  if (conv == 0) finish_range(po, po + L.c2_as);
  else           finish_range(po + L.c2_as, po + L.p_block_stride);
  if (has_lin0) finish_range(L.p_lin0_w, L.p_block0);
  if (has_lin1) finish_range(L.p_lin1_w, L.P);
  if (tid == 0) {
    const unsigned ncols = (unsigned)ncol;
    if (f.do_adam && !fault) {
      // counted by the column that finishes last; every column has read the count before it draws
      const unsigned long long done = atomicAdd(&f.step_counter[1], 1ULL);
      if (done == (unsigned long long)ncols - 1ULL) {
        f.step_counter[1] = 0ULL;
        atomicAdd(&f.step_counter[0], 1ULL);
      }
    }
    if (fault && f.final_launch) {                 // every column of this launch has read status[0] before the last ticket
      __threadfence();
      if (atomicAdd(f.status + 2, 1u) == ncols - 1u) {
        f.status[2] = 0u;
        f.status[1] += 1u;
        __hip_atomic_store(f.status, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// gatres_probe_xcd_dispatch: one workgroup per CU (the whole LDS, 1024 threads: the shape of the split launches), each records
// the XCD it runs on.
__global__ __launch_bounds__(1024) void xcd_probe_kernel(unsigned* __restrict__ out) {
  __shared__ unsigned char lds[LDS_BYTES];
  lds[threadIdx.x] = (unsigned char)threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = xcc_id() | ((unsigned)(lds[1] & 0u) << 8);
}

// The same bookkeeping after a split launch that no gatres_fused_finish follows (forward / inference launches)
__global__ __launch_bounds__(64) void fused_status_kernel(unsigned* __restrict__ status) {
  if (threadIdx.x == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
    status[1] += 1u;
    __hip_atomic_store(status, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Two split launches must never be in flight on one device at the same time: each needs its whole grid resident, and
// each could hold CUs the other waits for (the bounded spins would then poison both).  Launches on ONE stream are
// ordered anyway; when the stream changes, the new stream first waits for everything enqueued on the previous one.
// Skipped while `st` is being captured: the host that replays the graph calls gatres_fused_serialize before the replay.
static std::mutex g_split_mu;
static hipStream_t g_split_stream[64];
static bool g_split_seen[64] = {};
static int serialize_split_launch(hipStream_t st) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  std::lock_guard<std::mutex> lk(g_split_mu);
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (cs != hipStreamCaptureStatusNone) return 0;
  if (g_split_seen[dev] && g_split_stream[dev] != st) {
    hipStream_t prev = g_split_stream[dev];
    hipStreamCaptureStatus ps = hipStreamCaptureStatusNone;
    hipEvent_t ev;
    if (hipStreamIsCapturing(prev, &ps) == hipSuccess && ps == hipStreamCaptureStatusNone &&
        hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
      if (hipEventRecord(ev, prev) == hipSuccess) (void)hipStreamWaitEvent(st, ev, 0);
      (void)hipEventDestroy(ev);
    }
    (void)hipGetLastError();                     // (a stream the caller has destroyed meanwhile: nothing left to wait for)
  }
  // ... and for what the per-op backward may still have in flight on the library's side stream (its parameter-gradient
  // launches): they hold CUs a split launch needs resident
  if (gatres_side_t* side = gatres_side_peek()) {
    hipStreamCaptureStatus ss = hipStreamCaptureStatusNone;
    hipEvent_t ev;
    if (side->stream != st && hipStreamIsCapturing(side->stream, &ss) == hipSuccess && ss == hipStreamCaptureStatusNone &&
        hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
      if (hipEventRecord(ev, side->stream) == hipSuccess) (void)hipStreamWaitEvent(st, ev, 0);
      (void)hipEventDestroy(ev);
    }
    (void)hipGetLastError();
  }
  g_split_stream[dev] = st;
  g_split_seen[dev] = true;
  return 0;
}

static int fused_threads_small() {
  return gatres_knobs()->fused_threads;
}
static int threads_for(int nc) { return nc <= 32 ? fused_threads_small() : (nc == 64 ? 512 : 256); }

// CUs of the current device (cached): every workgroup of a split launch must be resident at once, one per CU.
static int device_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      cus = v;
    else
      cus = 1;               // unknown: never split
  }
  return cus;
}

// The window kernel applies when the plan knows the parts' row windows and both LDS layouts fit with them.
static bool window_kernel_fits(const Layout& L, const gatres_graph_t* g, int M, bool keep = false) {
  if (gatres_knobs()->fused_no_window || M < 2 || M > 8 || L.nc > 32 || threads_for(L.nc) != 1024) return false;
  const int k = M - 2;
  const int wr = g->window[k][0], ge = g->window[k][1], gm = g->window[k][2], hl = g->halo[k];
  if (wr <= 0 || L.xch_stride <= 0) return false;
  const int tiles = (g->max_segment_nodes + 15) / 16;
  const int ow = 16 * ((tiles + M - 1) / M);
  const long long kb = keep ? win_keep_bytes(L.nb, ow) : 0;
  return win_fwd_bytes(L.nc, wr, ow, ge, gm, hl) + kb <= LDS_BYTES &&
         win_bwd_bytes(L.nc, 1024, wr, ow, ge, gm, hl) + kb + (keep ? 4LL * ow * L.nc + 16 : 0) <= LDS_BYTES &&
         wr <= 65535 && ge <= 65535 && gm <= 65535;
}

// CUs per segment.  B = the co-residency bound of the layout (at most 8, whole grid resident, no more parts than 16-row
// tiles).  Preferred: B parts on the window kernel; the kept gradient tables then become parameter gradients on whatever
// CUs are left for consumer workgroups (batches below 32 snapshots) or in the stand-alone launch that follows
// (param_grads_stream_kernel).  Measured on gatres_small, C-Town, bs = 32 (B = 8; profiles/r02_split_sweep.txt), GPU time
// of the two launches: 8 parts 462 + 38 us; 6 parts + 2 consumers 526 us (the part that owns a segment's last 68 rows
// needs two trips through every conv1-width sparse stage and a fifth MFMA tile -- 384-node snapshots run 22 us faster --
// and the consumers finish 15 - 20 us after the parts); 7 + 1 607 us (ONE consumer streams an item at ~23 GB/s, the
// LDS-DMA rate of one CU beside 224 busy ones, and cannot keep up with two items per block).  GATRES_FUSED_PREFER_CONSUMERS=1
// restores round 2's earlier choice (B - 2 parts + two consumers).  Otherwise the whole-segment-table kernel at
// min(B, 4).  The choice depends on the plan only, never on the phases of a launch: the per-part hand-off epochs persist
// in scratch.  GATRES_FUSED_SPLIT=1..8 overrides.
static int fused_split(const Layout& L, const gatres_graph_t* g) {
  const int tiles = (g->max_segment_nodes + 15) / 16;
  const int padded = ((g->num_segments + 7) / 8) * 8;
  // Rounds: 49 .. 96 segments fit the chip only at 4 .. 2 parts each -- whole-segment tables, where a snapshot costs MORE
  // than in a batch of 32 (bs 64: 0.99 ms = 64.7 k snapshots/s against 75.5 k at bs 32).  The window kernel is launched twice
  // instead, each round with half the segments at the parts that fit then (hand-offs only ever cross the parts of ONE
  // segment; the parameter-gradient launch and the update follow the last round): C-Town, bs 56 / 64 / 72 / 80 / 88 / 96,
  // ms per step, resident vs rounds: profiles/r03_rounds.txt.  Three rounds of 32 measured slower than the resident choice
  // (bs 72 .. 96: 1.24 .. 1.29 vs 1.14 .. 1.27).  GATRES_FUSED_NO_ROUNDS=1: the resident choice.
  if (const int rp = rounds_parts_for(g->num_segments)) {
    const int per = ((padded / 2 + 7) / 8) * 8;
    if (L.split_max == rp && tiles >= rp && L.nb > 0 && per * rp <= device_cus() && !gatres_knobs()->fused_split &&
        !gatres_knobs()->fused_no_rounds && !gatres_knobs()->fused_prefer_consumers && window_kernel_fits(L, g, rp))
      return rp;
  }
  int B = L.split_max;
  while (B > 1 && (B > tiles || L.nb == 0 || padded * B > device_cus())) --B;   // (a partitioned / smaller device)
  if (const int v = gatres_knobs()->fused_split) {
    if (v >= 1 && v <= B) return v;
  }
  if (B >= 2 && window_kernel_fits(L, g, B) && !gatres_knobs()->fused_prefer_consumers) return B;
  if (B >= 4 && window_kernel_fits(L, g, B - 2) && !gatres_knobs()->fused_no_consumers) return B - 2;
  if (B >= 2 && window_kernel_fits(L, g, B)) return B;
  int m = 1;
  while (m * 2 <= B && m < 4) m *= 2;
  return m;
}

// Consumer workgroups per segment for the deferred parameter gradients: only in launches that run the backward
// phase, only with the 1024-thread kernel (the consumers reuse its LDS), and only if the whole grid -- per-snapshot
// workgroups plus consumers, one per CU -- is still resident at once.
static int fused_consumers(const Layout& L, const gatres_graph_t* g, int M) {
  if (gatres_knobs()->fused_no_consumers || L.nb == 0 || threads_for(L.nc) != 1024) return 0;
  const int padded = ((g->num_segments + 7) / 8) * 8;
  int c = (device_cus() - padded * M) / padded;
  const int cap = gatres_knobs()->fused_consumers_cap;   // default 2; measured: 1, 2 and 4 consumers per snapshot give the same step time
  if (c > cap) c = cap;
  return c > 0 ? c : 0;
}

static bool use_window_kernel(const FusedArgs& a, const gatres_graph_t* g) {
  return a.saved && window_kernel_fits(a.L, g, a.M);        // (it writes the saved tables: training launches only)
}

static int launch_fused(int nc, int threads, const FusedArgs& a0, const gatres_graph_t* g, hipStream_t st) {
  const bool window = threads == 1024 && nc <= 32 && use_window_kernel(a0, g);
  const bool cache = !gatres_knobs()->fused_nocache &&
                     cache_fits(nc, threads, g->max_segment_nodes, g->max_segment_edges_gat, g->max_segment_edges_mean);
  // every workgroup of a split launch must be resident at once: more segments than that go round by round
  int per_round = g->num_segments;
  if (a0.M > 1 && ((g->num_segments + 7) / 8) * 8 * (a0.M + a0.C) > device_cus()) {
    const int padded = ((g->num_segments + 7) / 8) * 8, fit = (device_cus() / a0.M) & ~7;
    const int rounds = fit > 0 ? (padded + fit - 1) / fit : 0;
    per_round = rounds > 0 ? (((padded + rounds - 1) / rounds + 7) / 8) * 8 : 0;        // (equal rounds: 72 segments = 40 + 32)
  }
  if (per_round <= 0) return GATRES_E_UNSUPPORTED;
  FusedArgs a = a0;
  for (int s0 = 0; s0 < g->num_segments; s0 += per_round) {
    a.seg0 = s0;
    a.seg_cnt = g->num_segments - s0 < per_round ? g->num_segments - s0 : per_round;
    const unsigned grid = (unsigned)(((a.seg_cnt + 7) / 8) * 8 * (a.M + a.C));
    const int rc = window ? gatres_fused_launch_window(&a, nc, grid, st)
                          : gatres_fused_launch_whole(&a, nc, threads, cache ? 1 : 0, grid, st);
    if (rc != 0) return rc;
  }
  return 0;
}
}  // namespace

// C-ABI ------------------------------------------------------------------------------------------------------

extern "C" int gatres_fused_supported(const gatres_model_t* m, const gatres_graph_t* g) {
  if (!m || !g || g->num_segments <= 0 || !g->seg_ptr) return 0;
  if (!(m->nc >= 4 && m->nc <= 128 && gatres_is_pow2(m->nc))) return 0;
  if (m->act_dtype != GATRES_DTYPE_F32) return 0;        // the per-snapshot kernels are fp32 (the 1e-5 parity path)
  // Wide models (gatres_large, nc = 128): the per-snapshot tables do not fit the LDS, the per-snapshot kernel would
  // run without them at 256 / 512 threads, and the per-op kernels (LDS-staged persistent projections, 256 slabs) are
  // then 1.6-1.8x faster on C-Town batches of 32 .. 128 snapshots.  GATRES_FUSED_WIDE=1 keeps the fused path for them.
  if (m->nc > 32 && !gatres_knobs()->fused_wide) return 0;
  if (g->max_segment_nodes > 4096) return 0;      // beyond this a snapshot should be spread over many CUs
  return nocache_fits(m->nc, threads_for(m->nc), g->max_segment_nodes, g->max_segment_edges_gat, g->max_segment_edges_mean)
             ? 1 : 0;
}

extern "C" int gatres_fused_cus_per_segment(const gatres_model_t* m, const gatres_graph_t* g) {
  Layout L;
  if (!gatres_fused_supported(m, g) || !make_layout_g(m, g, &L)) return 0;
  return fused_split(L, g);
}

// Zero the split-segment barrier state (epochs, XCD words, consumer counters, error word) of a scratch buffer: what a
// freshly zeroed buffer has.  Only needed after a launch was aborted or the buffer was handed over from elsewhere.
extern "C" int gatres_fused_reset_sync(const gatres_model_t* m, const gatres_graph_t* g, float* scratch, void* stream) {
  if (!m || !g || !scratch) return GATRES_E_BADARG;
  Layout L;
  if (!make_layout_g(m, g, &L)) return GATRES_E_UNSUPPORTED;
  if (fused_nodes_of(g) == 0) return 0;
  if (L.xch_stride > 0) {      // granule tags of an aborted run must not meet a restarted epoch count
    const hipError_t e = hipMemsetAsync(scratch + L.sc_xch, 0, (size_t)g->num_segments * L.xch_stride * 4, gatres_stream(stream));
    if (e != hipSuccess) return (int)e;
  }
  return (int)hipMemsetAsync(scratch + L.sc_flags, 0, (size_t)(L.flag_words + L.ready_words + L.col_words) * 4, gatres_stream(stream));
}

extern "C" int gatres_fused_window_kernel(const gatres_model_t* m, const gatres_graph_t* g) {
  Layout L;
  if (!gatres_fused_supported(m, g) || !make_layout_g(m, g, &L)) return 0;
  return window_kernel_fits(L, g, fused_split(L, g)) ? 1 : 0;
}

// Diagnostic: segment 0 of the next fused launches writes a 100 MHz wall-clock stamp at every stage boundary into
// stamps[0..capacity) (device memory).  Pass nullptr to switch it off.
extern "C" int gatres_fused_set_stamps(uint64_t* stamps, int32_t capacity) {
  g_stamps = reinterpret_cast<unsigned long long*>(stamps);
  g_stamp_cap = stamps ? capacity : 0;
  return 0;
}

extern "C" int gatres_fused_prepare_backward(const gatres_model_t* m, const gatres_graph_t* g, const float* params,
                                             float* scratch, void* stream) {
  if (!m || !g || !params || !scratch) return GATRES_E_BADARG;
  Layout L;
  if (!make_layout_g(m, g, &L)) return GATRES_E_UNSUPPORTED;
  return gatres_transpose_conv_weights(params, scratch + L.sc_wt, L.nb, L.nc, stream);
}

static std::atomic<int> g_xcd_rr[64];          // per device: 0 = not probed, 1 = no, 2 = workgroups 8 ids apart share an XCD
static int xcd_rr_cached() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  return g_xcd_rr[dev].load(std::memory_order_acquire) == 2 ? 1 : 0;
}
extern "C" int gatres_fused_run(const gatres_model_t* m, const gatres_graph_t* g, const float* params,
                                const float* x, const uint8_t* mask, const float* y, float* out, float* g_out,
                                float* loss_part, float* g_x, float* saved, float* scratch, int32_t phases,
                                void* stream) {
  if (!m || !g || !params || !x || !scratch) return GATRES_E_BADARG;
  if (!gatres_fused_supported(m, g)) return GATRES_E_UNSUPPORTED;
  FusedArgs a;
  if (!make_layout_g(m, g, &a.L)) return GATRES_E_UNSUPPORTED;
  if ((phases & GATRES_PHASE_FORWARD) && !out) return GATRES_E_BADARG;
  if ((phases & GATRES_PHASE_LOSS) && (!mask || !y || !out || !g_out || !loss_part)) return GATRES_E_BADARG;
  if ((phases & GATRES_PHASE_BACKWARD) && (!saved || !g_out)) return GATRES_E_BADARG;
  a.seg_ptr = g->seg_ptr;
  a.rowptr = g->rowptr; a.col = g->col; a.t_rowptr = g->t_rowptr; a.t_eid = g->t_eid; a.t_dst = g->t_dst;
  a.m_rowptr = g->m_rowptr; a.m_col = g->m_col; a.mt_rowptr = g->mt_rowptr; a.mt_dst = g->mt_dst;
  a.N = g->num_nodes;
  a.perm = g->perm;
  a.params = params; a.wt = scratch + a.L.sc_wt;
  a.x = x; a.mask = mask; a.y = y; a.out = out; a.g_out = g_out; a.loss_part = loss_part; a.g_x = g_x;
  a.saved = saved; a.scratch = scratch; a.slabs = scratch + a.L.sc_slabs;
  a.num_segments = g->num_segments;
  a.seg0 = 0; a.seg_cnt = g->num_segments;
  a.M = fused_split(a.L, g);
  a.safe_sync = gatres_knobs()->fused_safe_sync;
  // bits 1 / 2 (diagnostic build only, WRONG results): never-wait exchanges; dX epilogues without their global reads
  a.no_halo = gatres_knobs()->fused_no_halo | (gatres_knobs()->xch_nowait ? 2 : 0) | (gatres_knobs()->diag_nomask ? 4 : 0);
  a.C = (phases & GATRES_PHASE_BACKWARD) ? fused_consumers(a.L, g, a.M) : 0;
  a.sym = (g->flags & GATRES_GRAPH_SYMMETRIC) && !gatres_knobs()->fused_heartbeat ? 1 : 0;
  a.ptab = g->part_tables; a.ptab_m = g->part_tables_m; a.ptab_stride = g->part_tables_stride;
  {
    const int tiles = (g->max_segment_nodes + 15) / 16, part_rows = a.M > 0 ? 16 * ((tiles + a.M - 1) / a.M) : INT32_MAX;
    a.facts = ((g->flags & GATRES_GRAPH_DEG_LE6) ? 0x400 : 0) | (part_rows <= 64 ? 0x1000 : 0) |
              (((m->flags & GATRES_MODEL_INFERENCE) && phases == GATRES_PHASE_FORWARD) ? 0x2000 : 0) |
              ((xcd_rr_cached() && !gatres_knobs()->window_sync_start) ? 0x4000 : 0);
  }
  a.keep_lds = (phases & GATRES_PHASE_FORWARD) && (phases & GATRES_PHASE_BACKWARD) && !gatres_knobs()->fused_no_keep &&
           window_kernel_fits(a.L, g, a.M, true) ? 1 : 0;
  a.flags = reinterpret_cast<unsigned*>(scratch + a.L.sc_flags);
  a.err = reinterpret_cast<int*>(a.flags + a.L.flag_words - 32);
  a.ready = a.flags + a.L.flag_words;
  a.part_slabs = scratch + a.L.sc_part_slabs;
  a.SL = make_seg_layout(a.L.nb, a.L.nc, g->max_segment_nodes, g->max_segment_edges_gat);
  a.xch = reinterpret_cast<unsigned long long*>(scratch + a.L.sc_xch);
  a.urec = reinterpret_cast<int*>(scratch + a.L.sc_urec);
  a.XL = make_xch_layout(a.L.nc, g->max_segment_nodes, g->max_segment_edges_gat);
  a.stamps = g_stamps; a.stamp_cap = g_stamp_cap;
  a.phases = (phases & (GATRES_PHASE_FORWARD | GATRES_PHASE_BACKWARD)) | ((phases & GATRES_PHASE_LOSS) ? PH_LOSS : 0);
  hipStream_t st = gatres_stream(stream);
  if (a.M > 1) serialize_split_launch(st);
  int rc = launch_fused(m->nc, threads_for(m->nc), a, g, st);
  if (rc == 0 && a.M > 1 && !(phases & GATRES_PHASE_BACKWARD)) {      // (a backward launch is followed by gatres_fused_finish)
    hipLaunchKernelGGL(fused_status_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<unsigned*>(a.err));
    rc = gatres_launch_status();
  }
  return rc;
}

extern "C" int gatres_fused_serialize(void* stream) { return serialize_split_launch(gatres_stream(stream)); }

extern "C" int gatres_probe_xcd_dispatch(void* stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  const int have = g_xcd_rr[dev].load(std::memory_order_acquire);
  if (have != 0) return have == 2 ? 1 : 0;
  hipStream_t st = gatres_stream(stream);
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return 0;      // (not now)
  const int cus = device_cus();
  if (cus < 16 || cus > 4096) { g_xcd_rr[dev].store(1, std::memory_order_release); return 0; }
  unsigned* d = nullptr;
  if (hipMalloc(&d, sizeof(unsigned) * cus) != hipSuccess) return 0;
  std::vector<unsigned> h((size_t)cus, 0xffu);
  bool ok = hipMemsetAsync(d, 0xff, sizeof(unsigned) * cus, st) == hipSuccess;
  if (ok) {
    hipLaunchKernelGGL(xcd_probe_kernel, dim3((unsigned)cus), dim3(1024), 0, st, d);
    ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess &&
         hipMemcpy(h.data(), d, sizeof(unsigned) * cus, hipMemcpyDeviceToHost) == hipSuccess;
  }
  (void)hipFree(d);
  if (!ok) return 0;
  bool rr = true;
  for (int b = 0; b < cus && rr; ++b) rr = (h[b] & 0xfu) == (h[b & 7] & 0xfu) && h[b] != 0xffffffffu;
  g_xcd_rr[dev].store(rr ? 2 : 1, std::memory_order_release);
  return rr ? 1 : 0;
}

extern "C" int64_t gatres_fused_status_offset(const gatres_model_t* m, const gatres_graph_t* g) {
  Layout L;
  if (!gatres_fused_supported(m, g) || !make_layout_g(m, g, &L)) return -1;
  return L.sc_flags + L.flag_words - 32;
}

extern "C" __attribute__((visibility("hidden"))) int gatres_fused_param_grads_ex(const gatres_model_t* m, const gatres_graph_t* g,
                                                                                 const float* saved, float* scratch,
                                                                                 const uint64_t* step_counter, void* stream);
extern "C" int gatres_fused_param_grads(const gatres_model_t* m, const gatres_graph_t* g, const float* saved,
                                        float* scratch, void* stream) {
  return gatres_fused_param_grads_ex(m, g, saved, scratch, nullptr, stream);
}
// step_counter != null (gatres_train_step): the launch also snapshots the step count and the fault word for the update
// launch's sampling tail (status words 8 .. 11).  Returns 1 instead of 0 when it did (the caller may then ask for mask_next).
extern "C" __attribute__((visibility("hidden"))) int gatres_fused_param_grads_ex(const gatres_model_t* m, const gatres_graph_t* g,
                                                                                 const float* saved, float* scratch,
                                                                                 const uint64_t* step_counter, void* stream) {
  if (!m || !g || !saved || !scratch) return GATRES_E_BADARG;
  if (!gatres_fused_supported(m, g)) return GATRES_E_UNSUPPORTED;
  ParamGradArgs a;
  if (!make_layout_g(m, g, &a.L)) return GATRES_E_UNSUPPORTED;
  if (a.L.nb == 0) return 0;
  if (fused_consumers(a.L, g, fused_split(a.L, g)) > 0) return 0;     // done by the backward launch's consumers
  a.seg_ptr = g->seg_ptr; a.saved = saved; a.keep = scratch + a.L.sc_keep; a.slabs = scratch + a.L.sc_slabs;
  a.M = fused_split(a.L, g); a.part_slabs = scratch + a.L.sc_part_slabs;
  a.SL = make_seg_layout(a.L.nb, a.L.nc, g->max_segment_nodes, g->max_segment_edges_gat);
  a.wt = scratch + a.L.sc_wt;
  const dim3 grid((unsigned)(2 * a.L.nb * g->num_segments));
  hipStream_t st = gatres_stream(stream);
  if ((m->nc == 16 || m->nc == 32) && !gatres_knobs()->param_grads_no_stream) {
    if (step_counter) {
      unsigned* status = reinterpret_cast<unsigned*>(scratch + a.L.sc_flags + a.L.flag_words - 32);
      a.step_counter = reinterpret_cast<const unsigned long long*>(step_counter);
      a.status = fused_nodes_of(g) > 0 ? status : nullptr;
      a.snap = reinterpret_cast<unsigned long long*>(status + 8);
    }
#ifdef GATRES_PROBE_PG_DMA
    if (m->nc == 16) hipLaunchKernelGGL((param_grads_stream_kernel<16>), grid, dim3(PGS_THREADS), 0, st, a);
    else             hipLaunchKernelGGL((param_grads_stream_kernel<32>), grid, dim3(PGS_THREADS), 0, st, a);
#else
    if (m->nc == 16) hipLaunchKernelGGL((param_grads_reg_kernel<16>), grid, dim3(PGR_THREADS), 0, st, a);
    else             hipLaunchKernelGGL((param_grads_reg_kernel<32>), grid, dim3(PGR_THREADS), 0, st, a);
#endif
    const int rc = gatres_launch_status();
    return rc ? rc : (step_counter ? 1 : 0);
  }
  switch (m->nc) {
    case 4: hipLaunchKernelGGL((param_grads_kernel<4, 256>), grid, dim3(256), 0, st, a); break;
    case 8: hipLaunchKernelGGL((param_grads_kernel<8, 256>), grid, dim3(256), 0, st, a); break;
    case 16: hipLaunchKernelGGL((param_grads_kernel<16, 512>), grid, dim3(512), 0, st, a); break;
    case 32: hipLaunchKernelGGL((param_grads_kernel<32, 512>), grid, dim3(512), 0, st, a); break;
    case 64: hipLaunchKernelGGL((param_grads_kernel<64, 512>), grid, dim3(512), 0, st, a); break;
    case 128: hipLaunchKernelGGL((param_grads_kernel<128, 512>), grid, dim3(512), 0, st, a); break;
    default: return GATRES_E_UNSUPPORTED;
  }
  return gatres_launch_status();
}

// 1 when gatres_fused_param_grads_finish applies: streamed items (nc 16 / 32) in a launch of their own
static bool finish_folds(const gatres_model_t* m, const gatres_graph_t* g, const Layout& L) {
  if (!(m->nc == 16 || m->nc == 32) || gatres_knobs()->param_grads_no_stream || L.nb == 0) return false;
  return fused_consumers(L, g, fused_split(L, g)) == 0;
}

extern "C" int gatres_fused_finish_folds(const gatres_model_t* m, const gatres_graph_t* g) {
  Layout L;
  if (!gatres_fused_supported(m, g) || !make_layout_g(m, g, &L)) return 0;
  return finish_folds(m, g, L) ? 1 : 0;
}

extern "C" int gatres_fused_param_grads_finish(const gatres_model_t* m, const gatres_graph_t* g, const float* saved,
                                               float* scratch, float* grads, const float* loss_part, float* loss,
                                               int32_t do_adam, float* params, float* exp_avg, float* exp_avg_sq,
                                               uint64_t* step_counter, double lr, double beta1, double beta2, double eps,
                                               double weight_decay, const double* hparams, float grad_scale,
                                               int32_t block_lo, int32_t block_hi, void* stream) {
  if (!m || !g || !saved || !scratch || !grads) return GATRES_E_BADARG;
  if (do_adam && (!params || !exp_avg || !exp_avg_sq || !step_counter)) return GATRES_E_BADARG;
  if ((loss_part == nullptr) != (loss == nullptr)) return GATRES_E_BADARG;
  if (!gatres_fused_supported(m, g)) return GATRES_E_UNSUPPORTED;
  ParamGradArgs a;
  if (!make_layout_g(m, g, &a.L)) return GATRES_E_UNSUPPORTED;
  if (!finish_folds(m, g, a.L)) return GATRES_E_UNSUPPORTED;
  if (block_lo < 0 || block_hi > a.L.nb || block_lo >= block_hi) return GATRES_E_BADARG;
  if (do_adam && (block_lo != 0 || block_hi != a.L.nb)) return GATRES_E_BADARG;      // (the step is counted per launch)
  a.seg_ptr = g->seg_ptr; a.saved = saved; a.keep = scratch + a.L.sc_keep; a.slabs = scratch + a.L.sc_slabs;
  a.M = fused_split(a.L, g); a.part_slabs = scratch + a.L.sc_part_slabs;
  a.SL = make_seg_layout(a.L.nb, a.L.nc, g->max_segment_nodes, g->max_segment_edges_gat);
  a.wt = scratch + a.L.sc_wt;
  FinishArgs f;
  f.grads = grads; f.loss_part = loss_part; f.loss = loss; f.num_loss = g->num_segments * a.M; f.do_adam = do_adam ? 1 : 0;
  f.p = params; f.m = exp_avg; f.v = exp_avg_sq; f.step_counter = reinterpret_cast<unsigned long long*>(step_counter);
  f.lr = lr; f.b1 = beta1; f.b2 = beta2; f.eps = eps; f.wd = weight_decay; f.hp = hparams; f.grad_scale = grad_scale;
  f.wt = do_adam ? scratch + a.L.sc_wt : nullptr;
  f.status = fused_nodes_of(g) > 0 ? reinterpret_cast<unsigned*>(scratch + a.L.sc_flags + a.L.flag_words - 32) : nullptr;
  f.colcnt = reinterpret_cast<unsigned*>(scratch + a.L.sc_flags + a.L.flag_words + a.L.ready_words);
  f.S = g->num_segments; f.b_lo = block_lo; f.b_hi = block_hi; f.final_launch = block_lo == 0 ? 1 : 0;
  const dim3 grid((unsigned)(2 * (block_hi - block_lo) * g->num_segments));
  hipStream_t st = gatres_stream(stream);
  if (m->nc == 16) hipLaunchKernelGGL((param_grads_finish_kernel<16>), grid, dim3(PGR_THREADS), 0, st, a, f);
  else             hipLaunchKernelGGL((param_grads_finish_kernel<32>), grid, dim3(PGR_THREADS), 0, st, a, f);
  return gatres_launch_status();
}

extern "C" int gatres_fused_finish_hp(const gatres_model_t* m, const gatres_graph_t* g, float* scratch, float* grads,
                                      const float* loss_part, float* loss, int32_t do_adam, float* params,
                                      float* exp_avg, float* exp_avg_sq, uint64_t* step_counter, double lr, double beta1,
                                      double beta2, double eps, double weight_decay, const double* hparams,
                                      float grad_scale, const int32_t* mask_node_ptr, int32_t mask_graphs, double mask_rate,
                                      uint64_t mask_seed, uint8_t* mask_next, void* stream);

extern "C" __attribute__((visibility("hidden"))) int gatres_fused_finish_ex(
    const gatres_model_t* m, const gatres_graph_t* g, float* scratch, float* grads, const float* loss_part, float* loss,
    int32_t do_adam, float* params, float* exp_avg, float* exp_avg_sq, uint64_t* step_counter, double lr, double beta1,
    double beta2, double eps, double weight_decay, const double* hparams, float grad_scale, const int32_t* mask_node_ptr,
    int32_t mask_graphs, double mask_rate, uint64_t mask_seed, uint8_t* mask_next, int32_t use_snap, void* stream);

extern "C" int gatres_fused_finish(const gatres_model_t* m, const gatres_graph_t* g, float* scratch, float* grads,
                                   const float* loss_part, float* loss, int32_t do_adam, float* params,
                                   float* exp_avg, float* exp_avg_sq, uint64_t* step_counter, double lr, double beta1,
                                   double beta2, double eps, double weight_decay, float grad_scale, void* stream) {
  return gatres_fused_finish_hp(m, g, scratch, grads, loss_part, loss, do_adam, params, exp_avg, exp_avg_sq, step_counter,
                                lr, beta1, beta2, eps, weight_decay, nullptr, grad_scale, nullptr, 0, 0.0, 0, nullptr, stream);
}

extern "C" int gatres_fused_finish_hp(const gatres_model_t* m, const gatres_graph_t* g, float* scratch, float* grads,
                                      const float* loss_part, float* loss, int32_t do_adam, float* params,
                                      float* exp_avg, float* exp_avg_sq, uint64_t* step_counter, double lr, double beta1,
                                      double beta2, double eps, double weight_decay, const double* hparams,
                                      float grad_scale, const int32_t* mask_node_ptr, int32_t mask_graphs, double mask_rate,
                                      uint64_t mask_seed, uint8_t* mask_next, void* stream) {
  // (mask_next through the public entry: the caller vouches that gatres_fused_param_grads ran under gatres_train_step's
  //  sequence -- the snapshot it reads is only written there; without one the sampling tail is refused)
  if (mask_next) return GATRES_E_UNSUPPORTED;
  return gatres_fused_finish_ex(m, g, scratch, grads, loss_part, loss, do_adam, params, exp_avg, exp_avg_sq, step_counter, lr,
                                beta1, beta2, eps, weight_decay, hparams, grad_scale, mask_node_ptr, mask_graphs, mask_rate,
                                mask_seed, mask_next, 0, stream);
}

// (not part of include/gatres.h) use_snap: the parameter-gradient launch in front of this one (gatres_fused_param_grads_ex)
// left the step count / fault word snapshot: the update reads those instead of drawing tickets, and may sample the next mask
extern "C" __attribute__((visibility("hidden"))) int gatres_fused_finish_ex(
    const gatres_model_t* m, const gatres_graph_t* g, float* scratch, float* grads, const float* loss_part, float* loss,
    int32_t do_adam, float* params, float* exp_avg, float* exp_avg_sq, uint64_t* step_counter, double lr, double beta1,
    double beta2, double eps, double weight_decay, const double* hparams, float grad_scale, const int32_t* mask_node_ptr,
    int32_t mask_graphs, double mask_rate, uint64_t mask_seed, uint8_t* mask_next, int32_t use_snap, void* stream) {
  if (!m || !g || !scratch || !grads) return GATRES_E_BADARG;
  if (mask_next && !use_snap) return GATRES_E_BADARG;
  if (mask_next && (!do_adam || !mask_node_ptr || mask_graphs <= 0 || !(mask_rate >= 0.0 && mask_rate <= 1.0))) return GATRES_E_BADARG;
  if (do_adam && (!params || !exp_avg || !exp_avg_sq || !step_counter)) return GATRES_E_BADARG;
  if ((loss_part == nullptr) != (loss == nullptr)) return GATRES_E_BADARG;
  Layout L;
  if (!make_layout_g(m, g, &L)) return GATRES_E_UNSUPPORTED;
  const int reduce_blocks = (int)((L.P + 255) / 256);
  hipLaunchKernelGGL(reduce_adam_kernel, dim3((unsigned)(reduce_blocks + (mask_next ? mask_graphs * MASK_WGS_UPDATE : 0))), dim3(256), 0,
                     gatres_stream(stream),
                     scratch + L.sc_slabs, g->num_segments, g->num_segments * fused_split(L, g),
                     (long long)L.slab_stride, (long long)L.P, grads, loss_part, loss, do_adam, params, exp_avg, exp_avg_sq,
                     reinterpret_cast<unsigned long long*>(step_counter), lr, beta1, beta2, eps, weight_decay,
                     grad_scale, do_adam ? scratch + L.sc_wt : nullptr, L.nb, L.nc,
                     fused_nodes_of(g) > 0 ? reinterpret_cast<unsigned*>(scratch + L.sc_flags + L.flag_words - 32) : nullptr,
                     hparams, mask_node_ptr, mask_rate, (unsigned long long)mask_seed, mask_next,
                     reinterpret_cast<const unsigned long long*>(scratch + L.sc_flags + L.flag_words - 32 + 8),
                     (use_snap && do_adam) ? 1 : 0, reduce_blocks);
  return gatres_launch_status();
}

