// Window form of the fused per-snapshot kernel: the headline path (gatres_small, 8 parts per snapshot).
// Device code and commentary: k_fused_dev.h.
#include <climits>
#include "k_window_stages.h"

namespace {

// The waves that issued LDS-DMA in a stage (w0 and up; they own no MFMA tile there) wait for it to land before the stage's
// closing barrier.  (The sparse stages read LDS through asm statements, which hipcc does not treat as LDS reads: it no
// longer puts its own conservative s_waitcnt vmcnt(0) between a wave's DMA and that wave's next LDS access.)
__device__ __forceinline__ void dma_land(int w0) {
  if ((int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) >= w0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// n floats (n % 4 == 0, n <= 256) of small per-convolution vectors -> the tail of a W slot, by the last LDS-DMA wave
template <int THREADS>
__device__ __forceinline__ void vec_prefetch(float* dst, const float* __restrict__ src, int n) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave == THREADS / 64 - 1 && lane * 4 < n) __builtin_amdgcn_global_load_lds(src + lane * 4, dst, 16, 0, 0);
}
// A uniform value the compiler must take as new at this point: address arithmetic that depends on it cannot be hoisted above
// a wave-role branch (the MFMA stages split their waves into tile waves and LDS-DMA waves; hoisted, BOTH roles ran the other's
// scalar address chains and the v_readlane reloads of the spilled uniforms they need: 45 - 150 per stage of 130 - 300 VALU
// instructions, profiles/r03_kernel_resource_usage.txt).
__device__ __forceinline__ int launder_s(int v) {
  asm volatile("" : "+s"(v));
  return v;
}

// The kernel's arguments, re-read where they are used.  Layout / SegLayout / XchLayout hold ~60 offsets that the 15-block
// loops need a few at a time; read once at the top they are all live across both loops, which is most of what hipcc spills
// to VGPR lanes (a v_readlane per use, on the VALU port the stages are bound by).  A stage that opens with FRESH_ARGS()
// reads its offsets through a pointer to the kernarg segment that the compiler cannot see through (empty asm): scalar loads
// from the constant cache, dead again when the stage ends.
typedef const __attribute__((address_space(4))) FusedArgs* KArgs;
__device__ __forceinline__ KArgs kargs_fresh() {
  KArgs p = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}
#define FRESH_ARGS()                                                                                          \
  const KArgs ka_ = kargs_fresh();                                                                            \
  [[maybe_unused]] const auto& L = ka_->L; [[maybe_unused]] const auto& SL = ka_->SL; [[maybe_unused]] const auto& XL = ka_->XL

// The part's geometry, re-read where it is used.  The two phase prologues carve LDS into ~25 / ~45 tables whose addresses (and
// the part's row / edge ranges and list lengths) every stage of the block loops needs a handful of; held in SGPRs across the
// loops they were the rest of the spill reloads.  The prologue now writes them into this workgroup's record (FusedArgs::urec,
// GATRES_UREC_WORDS words: forward half | backward half) and a stage reads its fields back by scalar loads (FRESH_FWD /
// FRESH_BWD: same laundered-pointer idiom as FRESH_ARGS).  LDS addresses are byte offsets from the start of the LDS array
// (shifted views may be negative); NO_TABLE = the table does not exist in this launch (null pointer).
struct FwdRec {
  int lo, hi, ow, wr, oeg, elo, hcnt, ecnt;
  int xA, xB, wlA, wlB, hA, hB, hBw, y2T, al2L, sa2, sa1, sd2, sd1, fflag, hlist, elist, nbin, mbin, rp, colo, mrp, mcolo, mo1, mxin;
};
struct BwdRec {
  int lo, hi, ow, wr, wlo, elo, ewlo, weg, n0, hrcnt, hecnt, ercnt, eecnt;
  int RA, gpT, gy2T, ge2, ge1, gad2, gad1, hTw, hT2, hT1, asTw, asT2, asT1, adTo, adT2, adT1, alTw, alT2, alT1, xG2, xG1, gkeep,
      wlA, wlB, rp, colo, trp, teido, tdsto, mrp, mtrp, mtdsto, nbin, tout, mout, hrow, hedge, erow, eedge, bflag, mo1, mxin;
};
static_assert(sizeof(FwdRec) <= 4 * (GATRES_UREC_WORDS / 2) && sizeof(BwdRec) <= 4 * (GATRES_UREC_WORDS / 2), "record halves");
constexpr int NO_TABLE = INT_MIN;
template <class R>
__device__ __forceinline__ const __attribute__((address_space(4))) R* rec_fresh(const int* words) {
  auto p = (const __attribute__((address_space(4))) R*)(unsigned long long)words;
  asm volatile("" : "+s"(p));
  return p;
}
// after the prologue's stores and the barrier behind them: no stale line of the record in the scalar cache
__device__ __forceinline__ void rec_published() { asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory"); }
#define FRESH_FWD() const auto rf_ = rec_fresh<FwdRec>(myrec)
#define FRESH_BWD() const auto rb_ = rec_fresh<BwdRec>(myrec + GATRES_UREC_WORDS / 2)
#define LDS_TABLE(T, base, off) (reinterpret_cast<T*>((base) + (off)))
// A stage declares what it needs as a field list  LIST(T_, N_, I_)  = T_(type, table) ... N_(type, nullable table) ...
// I_(int field) ...  and opens with  REC_LOADS(LIST); REC_PINS(LIST, ...); REC_DEFS(LIST);  -- scalar loads of the fields,
// then the tables as pointers and the ints under their own names.  hipcc sinks every load to its first use, behind whatever
// VALU / SALU work of the stage's opening does not need it: that hides the scalar-cache round trip.  REC_PINS is an
// experiment switch (-DGATRES_REC_PINS): one empty asm that names every loaded value, so that all loads are issued and
// waited for at the stage's opening -- measured + 12 us per step (0.4154 - 0.4188 vs 0.4042 - 0.4062 ms, same box), and in
// front of the previous stage's closing barrier 0.4172 - 0.4182; one asm per value (a wait each: scalar loads return out of
// order) 0.4427.  profiles/r04_valu_probe.txt.
#define REC_LOAD_T(T, n) const int o_##n = RECP->n;
#define REC_LOAD_I(n) const int o_##n = RECP->n;
#define REC_PIN_T(T, n) , "s"(o_##n)
#define REC_PIN_I(n) , "s"(o_##n)
#define REC_DEF_T(T, n) [[maybe_unused]] T* n = LDS_TABLE(T, lds_raw, o_##n);
#define REC_DEF_N(T, n) [[maybe_unused]] T* n = o_##n == NO_TABLE ? nullptr : LDS_TABLE(T, lds_raw, o_##n);
#define REC_DEF_I(n) [[maybe_unused]] const int n = o_##n;
#define REC_LOADS(LIST) LIST(REC_LOAD_T, REC_LOAD_T, REC_LOAD_I)
#ifdef GATRES_REC_PINS
#define REC_PINS(LIST, ...) asm volatile("" ::"s"(0) LIST(REC_PIN_T, REC_PIN_T, REC_PIN_I) __VA_ARGS__)
#else
#define REC_PINS(LIST, ...) do {} while (0)
#endif
#define REC_DEFS(LIST) LIST(REC_DEF_T, REC_DEF_N, REC_DEF_I) [[maybe_unused]] const Rows rw = {lo, hi}
__device__ __forceinline__ u16* align16(u16* p) {
  return reinterpret_cast<u16*>((reinterpret_cast<uintptr_t>(p) + 15) & ~(uintptr_t)15);
}

// g_pre of the own rows (LDS [row][NC], shifted view) -> g_pre / max(in-degree in SimpleConv's graph, 1): the operand K3
// backward gathers (win_bwd_dst<MEAN, PRE>).  The dX1 stage of the kept-in-LDS form stores it that way itself (win_proj:
// cnt_rp); this pass serves lin1 backward's rows and the launches that run dX1 through seg_proj.
template <int NC, int THREADS>
__device__ __forceinline__ void scale_g_pre(float* gpT, const u16* mrp, int lo, int ow) {
  for (int idx = threadIdx.x; idx < ow * NC; idx += THREADS) {
    const int r = lo + idx / NC;
    const float cnt = (float)max((int)mrp[r + 1] - (int)mrp[r], 1);
    float* p = gpT + (size_t)r * NC + idx % NC;
    *p = *p / cnt;
  }
}

template <int NC, int THREADS, int PH = -1>
__global__ __launch_bounds__(THREADS) void gatres_window_kernel(const FusedArgs a) {
  // PH >= 0: the launch's phases and whether it keeps the ReLU sign masks in LDS are compile-time facts (separate register
  // allocations for the forward-only, backward-only and training instantiations); PH < 0: read from the arguments
  const int ph_ = PH < 0 ? a.phases : (PH & 0xff);
  const bool keep_ = PH < 0 ? a.keep_lds != 0 : (PH & 0x100) != 0;
  // PH & 0x200: a symmetric plan (no pacing granules) and no consumer workgroups in this launch (the common case: undirected
  // water networks, the parameter gradients as a launch of their own) -- the hand-offs' pace / drain paths are not compiled
  constexpr bool LEAN = PH >= 0 && (PH & 0x200) != 0;
  const int nC_ = LEAN ? 0 : a.C;
  // PH & 0x400: no row of any CSR of the plan has more than MAXD entries (GATRES_GRAPH_DEG_LE6): the stages' edge-at-a-time
  // paths and the alpha-table fallbacks are not compiled.  PH & 0x800: the plan carries this split's part tables (a record
  // that fails its magic check faults the launch instead of falling back to the in-kernel derivation).  PH & 0x1000: no part
  // owns more than 64 rows (the host knows from the largest segment and the split).
  constexpr bool NOHUB = PH >= 0 && (PH & 0x400) != 0;
  constexpr bool PTAB = PH >= 0 && (PH & 0x800) != 0;
  constexpr bool OW64 = PH >= 0 && (PH & 0x1000) != 0;
  // PH & 0x2000 (forward-only launches): nobody reads the saved activations (GATRES_MODEL_INFERENCE) -- their stores, a
  // tenth of the forward's VMEM and VALU work, are not compiled
  constexpr bool NOSAVE = PH >= 0 && (PH & 0x2000) != 0 && (PH & 0xff) == GATRES_PHASE_FORWARD;
  // PH & 0x4000: the host probed the device's dispatch (gatres_probe_xcd_dispatch): parts 8 ids apart share an XCD.  The launch
  // starts without its first cross-CU barrier and `local` is a compile-time truth -- the agent-scope forms of every granule
  // store and sweep are not compiled.  Every part still records its XCD id; the end of the launch compares (group_verify_local).
  constexpr bool XLOCAL = PH >= 0 && (PH & 0x4000) != 0;
  // Row padding of the MFMA stages' x operand tables (xA / xB forward, xG backward): 16 rows K floats apart share their banks,
  // K + 4 apart they do not (the x fragment reads of win_proj were 8- / 16-way conflicted).  Only where every writer and reader
  // of the table is a win_* stage of this instantiation: forward with the fallbacks compiled out (NOHUB), backward with
  // keep-in-LDS as a compile-time fact (its other form runs dX through seg_proj).
#ifdef GATRES_NO_XPAD          // (A/B builds: tests/micro/mk_probe.sh)
  constexpr int XPF = 0, XPB = 0;
#else
  constexpr int XPF = (NC == 32 && NOHUB) ? 4 : 0;
  constexpr int XPB = (NC == 32 && PH >= 0 && (PH & 0x100) != 0) ? 4 : 0;
#endif
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
  float* ldsf = reinterpret_cast<float*>(lds_raw);
  const Layout& L = a.L;
  const int M = a.M;
  if constexpr (!LEAN) {
    const int F = ((a.seg_cnt + 7) / 8) * 8 * M;
    if ((int)blockIdx.x >= F) {
      consumer_main<NC, THREADS>(a, (int)blockIdx.x - F, ldsf);
      return;
    }
  }
  const int within = blockIdx.x % (8 * M);
  const int seg = a.seg0 + (blockIdx.x / (8 * M)) * 8 + (within & 7), part = within >> 3;
  if (seg >= a.seg0 + a.seg_cnt) return;
  const int tid = threadIdx.x;
  const int wave_u = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Part tables of the plan (gatres_graph_t.part_tables, built once per topology by gatres_graph_part_tables_host): the
  // part's scalars come from the record's header -- ONE load per wave -- and its LDS tables are copied from the record
  // by LDS-DMA in the two phase prologues.  Without them (a plan that carries none, or tables for another split) both are
  // derived from the CSR arrays as in round 2: four barrier-separated passes with LDS atomics per phase.
  const int* pt = (PTAB || (a.ptab && a.ptab_m == M)) ? a.ptab + ((size_t)seg * M + part) * (size_t)a.ptab_stride : nullptr;
  int hv = 0;
  if (PTAB || pt) hv = pt[min((int)(threadIdx.x & 63), GATRES_PT_HEADER - 1)];
  auto H = [&](int field) { return __builtin_amdgcn_readlane(hv, field); };
#ifndef GATRES_NO_L2_TOUCH
  if constexpr (PTAB && NC == 32) {
    // While the header is on its way (an HBM-cold line: the previous launches of the step streamed 300 MB through the L2s): ask
    // for the lines the prologue's LDS-DMA will want NEXT -- the part's image, which follows the header in its record, and the
    // first block's W1 -- so that the second of two dependent cold reads is a hit.  Values unused (see the touch before lin1).
    if (ph_ & GATRES_PHASE_FORWARD) {
      if (tid < 64) { if (tid * 32 < a.ptab_stride) (void)*reinterpret_cast<const volatile int*>(pt + tid * 32); }
      else if (tid < 64 + 2 * NC * NC / 32 && L.nb > 0)
        (void)*reinterpret_cast<const volatile float*>(a.params + L.p_block0 + L.c1_W + (tid - 64) * 32);
    } else {
      for (int k = tid * 32; k < a.ptab_stride; k += THREADS * 32) (void)*reinterpret_cast<const volatile int*>(pt + k);
    }
  }
#endif
  int n0, n, e0, em0, t0, mt0;
  Rows rw;
  if (PTAB || pt) {
    if (H(GATRES_PT_MAGIC) != GATRES_PT_MAGIC_VALUE || H(GATRES_PT_M) != M) {       // (not this plan's tables)
      if (tid == 0) *a.err = 1;
      if constexpr (PTAB) return;              // (the partners' sweeps time out: the step is dropped, gatres_fused_finish)
      pt = nullptr;
    }
  }
  if (PTAB || pt) {
    n0 = H(GATRES_PT_N0); n = H(GATRES_PT_N); e0 = H(GATRES_PT_E0); em0 = H(GATRES_PT_EM0); t0 = H(GATRES_PT_T0);
    mt0 = H(GATRES_PT_MT0); rw.lo = H(GATRES_PT_LO); rw.hi = H(GATRES_PT_HI);
  } else {
    n0 = uni(a.seg_ptr[seg]); n = uni(a.seg_ptr[seg + 1]) - n0;
    e0 = uni(a.rowptr[n0]); em0 = uni(a.m_rowptr[n0]); t0 = uni(a.t_rowptr[n0]); mt0 = uni(a.mt_rowptr[n0]);
    const int tiles = (n + 15) >> 4;
    rw.lo = 16 * (int)((long long)tiles * part / M);
    rw.hi = min(n, 16 * (int)((long long)tiles * (part + 1) / M));
  }
  const int lo = rw.lo, ow = rw.hi - rw.lo;
  const bool ow64 = OW64 || ow <= 64;
  Group grp;
  grp.flags = a.flags + (size_t)seg * 8 * FLAG_STRIDE; grp.M = M; grp.part = part; grp.err = a.err;
  constexpr bool assume_local = XLOCAL;
  group_init<THREADS>(grp, XLOCAL);
  if constexpr (XLOCAL) grp.local = true;
  else if (a.safe_sync) grp.local = false;
  // granule exchange state of this part (epochs persist in word 2 of the part's flag line)
  const XchLayout& XL = a.XL;
  Xch xc;
  xc.base = a.xch + (size_t)seg * (size_t)(L.xch_stride / 2);
  xc.base_hb = xc.base + XL.hb;
  xc.M = M; xc.part = part; xc.err = a.err; xc.local = grp.local;
  xc.dead = GATRES_DIAG && (a.no_halo & 2) != 0;      // diagnostic (GATRES_XCH_NOWAIT=1, WRONG results): never wait for a partner -- what
                                       // the launch would take if every hand-off were free
  xc.ep = (unsigned)uni((int)__hip_atomic_load(grp.flags + part * FLAG_STRIDE + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  const XchBuf xbuf = xch_buffer(xc.base, XL.total);
  // pacing of the hand-offs by heartbeat granules: only when the plan does not promise two-sided halos; store drain at
  // every hand-off: only when consumer workgroups read this part's tables in the same launch (xch_after)
  const bool pace = LEAN ? false : !a.sym, drain = nC_ > 0;
  // first wave that issues LDS-DMA inside MFMA stages: the waves below it own a 16-row tile there (dma_copy16)
  constexpr int PW = 8;                    // waves that carry the work units of an MFMA stage (win_proj); NC == 32
  const int dw0 = NC == 32 ? PW : min((ow + 15) >> 4, THREADS / 64 - 4);
  constexpr int WL_FLOATS = 2 * NC * (2 * NC + 4) + 4 * NC;
  constexpr int WLB = (WL_FLOATS + 3) & ~3;
  // small vectors in the tails of the W slots (filled by LDS-DMA with the slot's matrix: no stage starts with a global load)
  constexpr int B1OFF = 2 * NC * (NC + 4) + 4 * NC;      // forward, slot A: conv1's bias behind W1 | att_src | att_dst
  constexpr int B2OFF = NC * (2 * NC + 4) + 2 * NC;      // forward, slot B: conv2's bias
  constexpr int A2OFF = 2 * NC * (NC + 4);               // backward, slot A: conv2's att_src | att_dst behind W2^T
  constexpr int A1OFF = NC * (2 * NC + 4);               // backward, slot B: conv1's att_src | att_dst behind W1^T
  static_assert(B1OFF + 2 * NC <= WLB && B2OFF + NC <= WLB && A2OFF + 2 * NC <= WLB && A1OFF + 4 * NC <= WLB, "W slot tails");
  const float* P = a.params;
  float* sc = a.scratch;
  // keep_: the top of LDS carries the ReLU sign masks from the forward to the backward phase of this launch
  unsigned char* lds_top = lds_raw + LDS_BYTES - (keep_ ? win_keep_bytes(L.nb, ow) : 0);
  unsigned long long* mo1 = keep_ ? reinterpret_cast<unsigned long long*>(lds_top) - lo : nullptr;   // [b * ow + row]
  unsigned* mxin = keep_ ? reinterpret_cast<unsigned*>(lds_top + 8LL * L.nb * ow) - lo : nullptr;
  [[maybe_unused]] int stamp_i = 0;
  STAMP();
  if (STAMPS_PTR && blockIdx.x == 0 && threadIdx.x == 0) STAMPS_PTR[a.stamp_cap] = clock64();

  // ---- the window: own rows and every row adjacent to them (in- and out-neighbours), as one contiguous range; edge ranges
  // (local ids = position - e0): own in-edges [elo, elo + oeg), window in-edges [ewlo, ewlo + weg), ...
  int wlo, whi, elo, oeg, ewlo, weg, melo, oem, tlo, otg, mtlo, otm;
  if (PTAB || pt) {
    wlo = H(GATRES_PT_WLO); whi = H(GATRES_PT_WHI); elo = H(GATRES_PT_ELO); oeg = H(GATRES_PT_OEG); ewlo = H(GATRES_PT_EWLO);
    weg = H(GATRES_PT_WEG); melo = H(GATRES_PT_MELO); oem = H(GATRES_PT_OEM); tlo = H(GATRES_PT_TLO); otg = H(GATRES_PT_OTG);
    mtlo = H(GATRES_PT_MTLO); otm = H(GATRES_PT_OTM);
  } else {
    int* mm = reinterpret_cast<int*>(lds_raw);
    if (tid == 0) { mm[0] = lo; mm[1] = rw.hi; }
    __syncthreads();
    int vmin = lo, vmax = rw.hi;
    for (int e = a.rowptr[n0 + lo] + tid; e < a.rowptr[n0 + rw.hi]; e += THREADS) {
      const int j = a.col[e] - n0; vmin = min(vmin, j); vmax = max(vmax, j + 1);
    }
    for (int t = a.t_rowptr[n0 + lo] + tid; t < a.t_rowptr[n0 + rw.hi]; t += THREADS) {
      const int j = a.t_dst[t] - n0; vmin = min(vmin, j); vmax = max(vmax, j + 1);
    }
    for (int e = a.m_rowptr[n0 + lo] + tid; e < a.m_rowptr[n0 + rw.hi]; e += THREADS) {
      const int j = a.m_col[e] - n0; vmin = min(vmin, j); vmax = max(vmax, j + 1);
    }
    for (int t = a.mt_rowptr[n0 + lo] + tid; t < a.mt_rowptr[n0 + rw.hi]; t += THREADS) {
      const int j = a.mt_dst[t] - n0; vmin = min(vmin, j); vmax = max(vmax, j + 1);
    }
    if (vmin < lo) atomicMin(&mm[0], vmin);
    if (vmax > rw.hi) atomicMax(&mm[1], vmax);
    __syncthreads();
    wlo = uni(mm[0]); whi = uni(mm[1]);
    __syncthreads();
    elo = uni(a.rowptr[n0 + lo]) - e0;        oeg = uni(a.rowptr[n0 + rw.hi]) - e0 - elo;
    ewlo = uni(a.rowptr[n0 + wlo]) - e0;      weg = uni(a.rowptr[n0 + whi]) - e0 - ewlo;
    melo = uni(a.m_rowptr[n0 + lo]) - em0;    oem = uni(a.m_rowptr[n0 + rw.hi]) - em0 - melo;
    tlo = uni(a.t_rowptr[n0 + lo]) - t0;      otg = uni(a.t_rowptr[n0 + rw.hi]) - t0 - tlo;
    mtlo = uni(a.mt_rowptr[n0 + lo]) - mt0;   otm = uni(a.mt_rowptr[n0 + rw.hi]) - mt0 - mtlo;
  }
  const int wr = whi - wlo;

  const SegLayout& SL = a.SL;
  float* segbase = a.saved + (int64_t)seg * SL.total;                 // training only: saved is never null here

  // Training launches (forward + loss + backward in one): the loss is formed where lin1 produces the predictions and handed
  // to lin1's backward through LDS -- the separate loss pass read out / y / mask back from global memory, and lin1 backward
  // then g_out and the saved final activation: two dependent L2 round trips at the turn of the launch.  Same arithmetic, same
  // summation order (thread t sums row lo + t, as the stand-alone pass): the same bits.
  constexpr int LB_R = THREADS / NC;               // row stride of lin1 backward's threads (two rows each at most)
  const bool fuse_loss = (ph_ & PH_LOSS) && (ph_ & GATRES_PHASE_FORWARD) && (ph_ & GATRES_PHASE_BACKWARD) &&
                         (OW64 || ow <= 2 * LB_R) && ow64 && (OW64 || ow <= THREADS / (NC / 4));
  float* dml = ldsf + 64;                          // [own row]: masked out - y, then g_out (inside the backward's `red` region)
  float xk[2] = {0.f, 0.f};                        // final activation of rows lo + rg, lo + rg + LB_R, column c (lin1 backward)

  int* const myrec = a.urec + (size_t)blockIdx.x * GATRES_UREC_WORDS;
  auto lds_off = [&](const void* p) { return (int)(reinterpret_cast<const unsigned char*>(p) - lds_raw); };

  if (ph_ & GATRES_PHASE_FORWARD) {
    constexpr bool xreg = NC == 32;
    {  // ---- prologue: LDS carve-up, the part's tables, lin0; the geometry goes into the record
    // LDS: [hA wr x 2NC | hB wr x NC | sa wr x 2 | sd own x 2 | xA own x NC | xB own x 2NC | W slot A | W slot B] topology
    float* hAw = ldsf;
    float* hBw = hAw + (size_t)wr * 2 * NC;
    float* saw = hBw + (size_t)wr * NC;
    float* sdo = saw + (size_t)((wr * 2 + 3) & ~3);      // (every table starts 16-byte aligned: ds_read_b128)
    float* xAo = sdo + (size_t)((ow * 2 + 3) & ~3);
    float* xBo = xAo + (size_t)ow * (NC + XPF);
    float* wlA = reinterpret_cast<float*>(lds_raw + ((reinterpret_cast<unsigned char*>(xBo + (size_t)ow * (2 * NC + XPF)) - lds_raw + 15) & ~15));
    float* wlB = wlA + WLB;
    u16* tp = reinterpret_cast<u16*>(wlB + WLB);
    u16* rpo = tp;             tp += even(ow + 1);
    u16* colo = tp;            tp += even(oeg);
    u16* mrpo = tp;            tp += even(ow + 1);
    u16* mcolo = tp;           tp += even(oem);
    tp = align16(tp);
    u16* nbin = tp;            tp += 8 * ow;       // padded in-edge descriptors of the own rows (k_window_stages.h)
    u16* mbin = tp;            tp += 8 * ow;
    int* hcounter = reinterpret_cast<int*>(tp);
    const u16* f_img_end = tp;
    u16* hlist = tp + 2;                 // import list: remote sources of own in-edges
    // export flags of the own rows (one byte each, from the export list; the top of the list space): the stages that
    // produce exported rows store their granules themselves (XOut, k_window_stages.h) -- for NC == 32, where every
    // producer is a win_* stage
    unsigned char* fflag_o = lds_top - ((ow + 15) & ~15);
    const int hcap = max(0, (int)((fflag_o - reinterpret_cast<unsigned char*>(hlist)) / 4)) & ~1;
    u16* elist = hlist + hcap;           // export list: own rows some partner's row has an in-edge from
    const unsigned char* fflag = fflag_o - lo;
    // index-shifted views: absolute local row / relative own-edge indices work unchanged in the stage functions
    float* hA = hAw - wlo * 2 * NC;
    float* hB = hBw - wlo * NC;
    float* sa2 = saw - wlo * 2;  float* sa1 = saw - wlo;                 // a_src tables (H = 2 / H = 1)
    float* sd2 = sdo - lo * 2;   float* sd1 = sdo - lo;
    float* xA = xAo - lo * (NC + XPF);
    float* xB = xBo - lo * (2 * NC + XPF);
    const u16* rp = rpo - lo;  const u16* mrp = mrpo - lo;
    int hcnt = 0, ecnt = 0;
    if (PTAB || pt) {
      // the record's forward image is this LDS range, verbatim; the two hand-off lists follow it in the record
      hcnt = H(GATRES_PT_F_HCNT); ecnt = H(GATRES_PT_F_ECNT);
      if (H(GATRES_PT_F_IMG_WORDS) * 2 != (int)(f_img_end - rpo) || hcnt > hcap || ecnt > hcap) {
        if (tid == 0) *a.err = 1;
        hcnt = min(hcnt, hcap); ecnt = min(ecnt, hcap);
      } else {
        dma_copy4<THREADS>(reinterpret_cast<float*>(rpo), reinterpret_cast<const float*>(pt + H(GATRES_PT_F_IMG)),
                           H(GATRES_PT_F_IMG_WORDS), 0);
      }
      dma_copy4<THREADS>(reinterpret_cast<float*>(hlist), reinterpret_cast<const float*>(pt + H(GATRES_PT_F_HLIST)), (hcnt + 1) / 2, 0);
      dma_copy4<THREADS>(reinterpret_cast<float*>(elist), reinterpret_cast<const float*>(pt + H(GATRES_PT_F_ELIST)), (ecnt + 1) / 2, 0);
    } else {
      copy_rowptr16<THREADS>(rpo, a.rowptr, n0 + lo, ow, e0 + elo);
      copy_idx16<THREADS>(colo, a.col, e0 + elo, oeg, n0);
      copy_rowptr16<THREADS>(mrpo, a.m_rowptr, n0 + lo, ow, em0 + melo);
      copy_idx16<THREADS>(mcolo, a.m_col, em0 + melo, oem, n0);
    }
    for (int k = tid; k < ((ow + 15) & ~15) / 4; k += THREADS) reinterpret_cast<unsigned*>(fflag_o)[k] = 0u;
    float* xcur = segbase + SL.xin;
    if (L.nb > 0) {
      const float* pb0 = P + L.p_block0;
      w_prefetch<NC, 2 * NC, EPI_ATT, THREADS>(wlA, pb0 + L.c1_W, pb0 + L.c1_as, pb0 + L.c1_ad, 0);
      vec_prefetch<THREADS>(wlA + B1OFF, pb0 + L.c1_b, 2 * NC);
    }
    {  // lin0 (+ the caller-side x[mask] = 0)
      const float* w = P + L.p_lin0_w;
      const float* b = P + L.p_lin0_b;
      for (int idx = rw.lo * (NC / 4) + tid; idx < rw.hi * (NC / 4); idx += THREADS) {
        const int r = idx / (NC / 4), c0 = (idx % (NC / 4)) * 4;
        const int node = ext_id(a.perm, n0 + r);
        const float xv = (a.mask && a.mask[node]) ? 0.f : a.x[node];
        // (keep-in-LDS launches: the masked input of the own rows rides in block 0's slot of the ReLU-mask table, which no block
        //  uses -- lin0 backward, the launch's last step, then needs no perm -> mask -> x chain of cold global reads)
#ifndef GATRES_NO_XKEEP
        if (mxin && L.nb > 0 && c0 == 0) mxin[r] = __float_as_uint(xv);
#endif
        const float4 wv = ld4(w + c0), bv = ld4(b + c0);
        float4 o;
        o.x = xv * wv.x + bv.x; o.y = xv * wv.y + bv.y; o.z = xv * wv.z + bv.z; o.w = xv * wv.w + bv.w;
        if (!NOSAVE) st4(xcur + (unsigned)(r * NC + c0), o);
        st4(xA + (unsigned)(r * (NC + XPF) + c0), o);
      }
    }
    dma_land(0);
    __syncthreads();
    if (!PTAB && !pt) {
      build_nbr_in<THREADS>(rw, rp, colo, oeg, false, nbin);
      build_nbr_in<THREADS>(rw, mrp, mcolo, oem, true, mbin);
      hcnt = uni(build_halo<THREADS>(rp, colo, nullptr, rw, hlist, nullptr, hcap, hcounter));
      ecnt = uni(build_export_rows<THREADS>(a.t_rowptr, a.t_dst, n0, rw, elist, hcap, hcounter));
      if (hcnt > hcap || ecnt > hcap) {    // (the host sizes the lists from gatres_graph_t.halo: cannot happen with a sane plan)
        if (tid == 0) *a.err = 1;
        hcnt = min(hcnt, hcap); ecnt = min(ecnt, hcap);
      }
    }
    if (xreg)
      for (int k = tid; k < ecnt; k += THREADS) fflag_o[(int)elist[k] - lo] = 1;
    if (tid == 0) {
      FwdRec r;
      r.lo = rw.lo; r.hi = rw.hi; r.ow = ow; r.wr = wr; r.oeg = oeg; r.elo = elo; r.hcnt = hcnt; r.ecnt = ecnt;
      r.xA = lds_off(xA); r.xB = lds_off(xB); r.wlA = lds_off(wlA); r.wlB = lds_off(wlB); r.hA = lds_off(hA); r.hB = lds_off(hB);
      r.hBw = lds_off(hBw); r.y2T = lds_off(hAw - wlo * NC); r.al2L = lds_off(hAw + (size_t)wr * NC);
      r.sa2 = lds_off(sa2); r.sa1 = lds_off(sa1); r.sd2 = lds_off(sd2); r.sd1 = lds_off(sd1); r.fflag = lds_off(fflag);
      r.hlist = lds_off(hlist); r.elist = lds_off(elist); r.nbin = lds_off(nbin); r.mbin = lds_off(mbin);
      r.rp = lds_off(rp); r.colo = lds_off(colo); r.mrp = lds_off(mrp); r.mcolo = lds_off(mcolo);
      r.mo1 = mo1 ? lds_off(mo1) : NO_TABLE; r.mxin = mxin ? lds_off(mxin) : NO_TABLE;
      const int* src = reinterpret_cast<const int*>(&r);
#pragma unroll
      for (int k = 0; k < (int)(sizeof(FwdRec) / 4); ++k)
        __hip_atomic_store(myrec + k, src[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // the barrier below is `s_waitcnt lgkmcnt(0); s_barrier` on gfx950 -- no vmcnt: drain the record's stores here, or the
      // other waves' scalar loads (rec_published) may still find the previous launch's record in L2
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    }  // ---- (end of the prologue's scope: the block loop below sees the record only)
    __syncthreads();           // export flags and the record are published
    rec_published();
    auto xout = [&](bool on, const unsigned char* flag, long long t_rows, long long t_small) {
      XOut x;
      x.xb = xbuf; x.ep = xc.ep + 1u; x.local = xc.local; x.on = on; x.flag = flag;
      x.t_rows = (unsigned)t_rows; x.t_small = (unsigned)t_small;
      return x;
    };
    STAMP();
#define RECP rf_
    const int nb = L.nb;
    for (int b = 0; b < nb; ++b) {
      // (per stage: FRESH_ARGS() / FRESH_FWD() re-read the offsets and tables the stage needs; base / pb are formed where
      // they are used)
      {
        FRESH_ARGS(); FRESH_FWD();
#define LIST_(T_, N_, I_) I_(lo) I_(hi) T_(float, xA) T_(float, wlA) T_(float, wlB) T_(float, hA) T_(float, sa2) T_(float, sd2) T_(const unsigned char, fflag)
        REC_LOADS(LIST_);
        [[maybe_unused]] const float* pb = P + L.p_block0 + (int64_t)launder_s(b) * L.p_block_stride;
        [[maybe_unused]] float* base = segbase + (int64_t)launder_s(b) * SL.bstride;
        REC_PINS(LIST_, , "s"(pb), "s"(base));
        REC_DEFS(LIST_);
#undef LIST_
        // LDS-DMA rides on the MFMA stages, issued by their tile-less waves: W2 | att | bias of this block while proj1 runs,
        // W1 | att | bias of the next block while proj2 runs
        if (NC != 32 || wave_u >= dw0) {           // (NC == 32: the LDS-DMA waves; their address chains stay in here)
          w_prefetch<2 * NC, NC, EPI_ATT, THREADS>(wlB, pb + L.c2_W, pb + L.c2_as, pb + L.c2_ad, dw0);
          vec_prefetch<THREADS>(wlB + B2OFF, pb + L.c2_b, NC);
        }
        if constexpr (NC == 32) {
          if (wave_u < PW) {                         // the tile waves
            win_proj<NC, 2 * NC, 2, EPI_ATT, 2, PW, THREADS, NC + XPF>(rw, xA, wlA, NOSAVE ? nullptr : base + SL.h1, 0, hA,
                                                         NOSAVE ? nullptr : base + SL.as1, NOSAVE ? nullptr : base + SL.ad1, sa2,
                                                         sd2, nullptr, nullptr, nullptr, nullptr, xout(xreg, fflag, XL.f1h, XL.f1a));
          }
        } else
          seg_proj<NC, 2 * NC, 2, EPI_ATT, THREADS, true, true>(rw, xA, 0, pb + L.c1_W, base + SL.h1, 0, hA, 0, pb + L.c1_as,
                                                                 pb + L.c1_ad, base + SL.as1, base + SL.ad1, 0, sa2, sd2,
                                                                 nullptr, 0, nullptr, 0, wlA);
#ifndef GATRES_PROBE_NO_FWD_LAND      // (timing probe, WRONG results: the forward's projection stages without the landing wait of the next W)
        dma_land(dw0);
#endif
        lds_barrier();                                      // own rows of h1 / a_src are in LDS; the saved copies drain meanwhile
      }
      {
        FRESH_ARGS(); FRESH_FWD();
#define LIST_(T_, N_, I_) I_(lo) I_(hi) T_(float, hA) T_(float, sa2) T_(const u16, hlist) T_(const u16, elist) I_(hcnt) I_(ecnt)
        REC_LOADS(LIST_);
        REC_PINS(LIST_);
        REC_DEFS(LIST_);
#undef LIST_
        ++xc.ep;                                    // exchange F1: the gathers below read h1 / a_src of neighbour rows
        if (!xreg) xch_export2<2 * NC, 2, THREADS>(xc, xbuf, elist, ecnt, hA, (unsigned)XL.f1h, elist, ecnt, sa2, (unsigned)XL.f1a);
        xch_import2<2 * NC, 2, THREADS>(xc, xbuf, hlist, hcnt, (unsigned)XL.f1h, hA, hlist, hcnt, (unsigned)XL.f1a, sa2);
        xch_after<THREADS>(xc, pace, drain);
        lds_barrier();
        STAMP();
      }
      {
        FRESH_ARGS(); FRESH_FWD();
#define LIST_(T_, N_, I_) I_(lo) I_(hi) T_(const u16, nbin) T_(const u16, rp) T_(const u16, colo) T_(float, hA) T_(float, sa2) T_(float, sd2) T_(float, hBw) T_(float, wlA) T_(float, xB) T_(const unsigned char, fflag) N_(unsigned long long, mo1) I_(ow) I_(oeg) I_(wr) I_(elo)
        REC_LOADS(LIST_);
        [[maybe_unused]] const float* pb = P + L.p_block0 + (int64_t)launder_s(b) * L.p_block_stride;
        [[maybe_unused]] float* base = segbase + (int64_t)launder_s(b) * SL.bstride;
        REC_PINS(LIST_, , "s"(pb), "s"(base));
        REC_DEFS(LIST_);
#undef LIST_
        // K2 conv1: alpha -> HBM, the gather's o1 -> HBM + the x buffer of proj2
        if (NOHUB || __builtin_expect(2 * oeg <= wr * NC, 1)) {
          if constexpr (NC == 32) {
            win_fwd_agg<true, 2, NC, THREADS, NOHUB, 2 * NC + XPF>(rw, nbin, rp, colo, hA, sa2, sd2, NOSAVE ? nullptr : base + SL.al1, elo, hBw, wlA + B1OFF,
                                              NOSAVE ? nullptr : base + SL.o1, 0, xB, mo1 ? mo1 + b * ow : nullptr, xout(false, fflag, 0, 0));
          } else {
            win_softmax<2, THREADS>(rw, nbin, rp, colo, sa2, sd2, base + SL.al1, elo, hBw);
            lds_barrier();
            win_gather<true, 2, NC, THREADS>(rw, nbin, rp, colo, hA, hBw, wlA + B1OFF, base + SL.o1, 0, xB,
                                             mo1 ? mo1 + b * ow : nullptr);
          }
        } else {
          dma_land(dw0);                            // (every wave reads LDS below)
          seg_softmax<2, false, THREADS>(rw, rp, colo, sa2, sd2, 0, base + SL.al1, elo, nullptr);
          __syncthreads();
          seg_gather<true, 2, NC, THREADS, 1>(rw, rp, colo, hA, 0, base + SL.al1, elo, pb + L.c1_b, base + SL.o1, 0, xB, 0,
                                               mo1 ? mo1 + b * ow : nullptr);
        }
        lds_barrier();
        STAMP();
      }
      {
        FRESH_ARGS(); FRESH_FWD();
#define LIST_(T_, N_, I_) I_(lo) I_(hi) T_(float, xB) T_(float, wlA) T_(float, wlB) T_(float, hB) T_(float, sa1) T_(float, sd1) T_(const unsigned char, fflag)
        REC_LOADS(LIST_);
        [[maybe_unused]] const float* pb = P + L.p_block0 + (int64_t)launder_s(b) * L.p_block_stride;
        [[maybe_unused]] float* base = segbase + (int64_t)launder_s(b) * SL.bstride;
        REC_PINS(LIST_, , "s"(pb), "s"(base));
        REC_DEFS(LIST_);
#undef LIST_
        if (b + 1 < nb && (NC != 32 || wave_u >= dw0)) {
          const float* pn = pb + L.p_block_stride;
          w_prefetch<NC, 2 * NC, EPI_ATT, THREADS>(wlA, pn + L.c1_W, pn + L.c1_as, pn + L.c1_ad, dw0);
          vec_prefetch<THREADS>(wlA + B1OFF, pn + L.c1_b, 2 * NC);
        }
        if constexpr (NC == 32) {
          if (wave_u < PW) {
            win_proj<2 * NC, NC, 1, EPI_ATT, 2, PW, THREADS, 2 * NC + XPF>(rw, xB, wlB, NOSAVE ? nullptr : base + SL.h2, 0, hB,
                                                         NOSAVE ? nullptr : base + SL.as2, NOSAVE ? nullptr : base + SL.ad2, sa1,
                                                         sd1, nullptr, nullptr, nullptr, nullptr, xout(xreg, fflag, XL.f2h, XL.f2a));
          }
        } else
          seg_proj<2 * NC, NC, 1, EPI_ATT, THREADS, true, true>(rw, xB, 0, pb + L.c2_W, base + SL.h2, 0, hB, 0, pb + L.c2_as,
                                                               pb + L.c2_ad, base + SL.as2, base + SL.ad2, 0, sa1, sd1,
                                                               nullptr, 0, nullptr, 0, wlB);
#ifndef GATRES_PROBE_NO_FWD_LAND
        dma_land(dw0);
#endif
        lds_barrier();
      }
      {
        FRESH_ARGS(); FRESH_FWD();
#define LIST_(T_, N_, I_) I_(lo) I_(hi) T_(float, hB) T_(float, sa1) T_(const u16, hlist) T_(const u16, elist) I_(hcnt) I_(ecnt)
        REC_LOADS(LIST_);
        REC_PINS(LIST_);
        REC_DEFS(LIST_);
#undef LIST_
        ++xc.ep;                                    // exchange F2
        if (!xreg) xch_export2<NC, 1, THREADS>(xc, xbuf, elist, ecnt, hB, (unsigned)XL.f2h, elist, ecnt, sa1, (unsigned)XL.f2a);
        xch_import2<NC, 1, THREADS>(xc, xbuf, hlist, hcnt, (unsigned)XL.f2h, hB, hlist, hcnt, (unsigned)XL.f2a, sa1);
        xch_after<THREADS>(xc, pace, drain);
        lds_barrier();
        STAMP();
      }
      // K2 conv2: y2 in the lower half of the h1 window (its upper half: the alpha table of rows beyond the slot path)
      {
        FRESH_ARGS(); FRESH_FWD();
#define LIST_(T_, N_, I_) I_(lo) I_(hi) T_(const u16, nbin) T_(const u16, rp) T_(const u16, colo) T_(float, hB) T_(float, sa1) T_(float, sd1) T_(float, wlB) T_(float, y2T) T_(float, al2L) T_(const unsigned char, fflag) I_(oeg) I_(wr) I_(elo)
        REC_LOADS(LIST_);
        [[maybe_unused]] const float* pb = P + L.p_block0 + (int64_t)launder_s(b) * L.p_block_stride;
        [[maybe_unused]] float* base = segbase + (int64_t)launder_s(b) * SL.bstride;
        REC_PINS(LIST_, , "s"(pb), "s"(base));
        REC_DEFS(LIST_);
#undef LIST_
        if (NOHUB || __builtin_expect(oeg <= wr * NC, 1)) {
          if constexpr (NC == 32) {
            win_fwd_agg<false, 1, NC, THREADS, NOHUB>(rw, nbin, rp, colo, hB, sa1, sd1, NOSAVE ? nullptr : base + SL.al2, elo, al2L, wlB + B2OFF, y2T,
                                               0, nullptr, nullptr, xout(xreg, fflag, XL.f3, 0));
          } else {
            win_softmax<1, THREADS>(rw, nbin, rp, colo, sa1, sd1, base + SL.al2, elo, al2L);
            lds_barrier();
            win_gather<false, 1, NC, THREADS>(rw, nbin, rp, colo, hB, al2L, wlB + B2OFF, y2T, 0, nullptr, nullptr);
          }
        } else {
          dma_land(dw0);
          seg_softmax<1, false, THREADS>(rw, rp, colo, sa1, sd1, 0, base + SL.al2, elo, nullptr);
          __syncthreads();
          seg_gather<false, 1, NC, THREADS, 1>(rw, rp, colo, hB, 0, base + SL.al2, elo, pb + L.c2_b, y2T, 0);
        }
        lds_barrier();
      }
      {
        FRESH_ARGS(); FRESH_FWD();
#define LIST_(T_, N_, I_) I_(lo) I_(hi) T_(float, y2T) T_(const u16, hlist) T_(const u16, elist) I_(hcnt) I_(ecnt) I_(oeg) I_(wr)
        REC_LOADS(LIST_);
        REC_PINS(LIST_);
        REC_DEFS(LIST_);
#undef LIST_
        ++xc.ep;                                    // exchange F3: K3 averages y2 over neighbour rows
        if (!(xreg && oeg <= wr * NC)) xch_export2<NC, 0, THREADS>(xc, xbuf, elist, ecnt, y2T, (unsigned)XL.f3, elist, 0, y2T, 0u);
        xch_import2<NC, 0, THREADS>(xc, xbuf, hlist, hcnt, (unsigned)XL.f3, y2T, hlist, 0, 0u, y2T);
        xch_after<THREADS>(xc, pace, drain);
        lds_barrier();
        STAMP();
      }
      {
        FRESH_ARGS(); FRESH_FWD();
#define LIST_(T_, N_, I_) I_(lo) I_(hi) T_(const u16, mbin) T_(const u16, mrp) T_(const u16, mcolo) T_(float, y2T) T_(float, xA) N_(unsigned, mxin) I_(ow)
        REC_LOADS(LIST_);
        REC_PINS(LIST_);
        REC_DEFS(LIST_);
#undef LIST_
        // K3: residual from the x buffer, result back into it (and to HBM: saved xin of the next block)
        float* xnext = segbase + (int64_t)(launder_s(b) + 1) * SL.bstride + SL.xin;
        win_mean_fwd<NC, THREADS, NOHUB, NC + XPF>(rw, mbin, mrp, mcolo, y2T, xA, NOSAVE ? nullptr : xnext, xA,
                                       (mxin && b + 1 < nb) ? mxin + (b + 1) * ow : nullptr);
        lds_barrier();
        STAMP();
      }
    }
    // (assume_local: every partner has exchanged rows with this part by now -- its XCD id is on record; checked here, off the
    //  launch's tail, in a launch without a backward phase)
    if (assume_local && !(ph_ & GATRES_PHASE_BACKWARD)) group_verify_local(grp);
#ifndef GATRES_NO_L2_TOUCH
    if constexpr (PTAB && NC == 32) {
      // Training launches: the backward prologue opens with LDS-DMA of tables nobody has touched for a whole forward phase -- the
      // part's backward image and hand-off lists (the tail of its record) and the last block's W2^T | att: HBM-cold, and every CU
      // asks at the same moment.  One load per 128 bytes here, under lin1 and the loss, brings the lines into this XCD's L2; the
      // values are not used (volatile loads: the compiler counts them, nothing waits for them before the prologue's own waits).
      if (ph_ & GATRES_PHASE_BACKWARD) {
        const int b0 = H(GATRES_PT_B_IMG);
        for (int k = b0 + tid * 32; k < a.ptab_stride; k += THREADS * 32) (void)*reinterpret_cast<const volatile int*>(pt + k);
        if (L.nb > 0 && tid >= THREADS / 2) {
          const int t2 = tid - THREADS / 2;
          constexpr int WT_LINES = 2 * NC * NC / 32;                     // W2^T: 2NC x NC floats
          const float* wt2 = a.wt + (int64_t)(L.nb - 1) * 4 * NC * NC + 2 * NC * NC;
          if (t2 < WT_LINES) (void)*reinterpret_cast<const volatile float*>(wt2 + t2 * 32);
          else if (t2 < WT_LINES + 2)
            (void)*reinterpret_cast<const volatile float*>(P + L.p_block0 + (int64_t)(L.nb - 1) * L.p_block_stride + L.c2_as + (t2 - WT_LINES) * 32);
        }
      }
    }
#endif
    FRESH_FWD();
#define LIST_(T_, N_, I_) I_(lo) I_(hi) T_(float, xA)
    REC_LOADS(LIST_);
    REC_DEFS(LIST_);
#undef LIST_
    {  // lin1
      constexpr int G = NC / 4;
      const float4 wv = ld4(P + L.p_lin1_w + (tid % G) * 4);
      const float bias = P[L.p_lin1_b];
      const int rounds = (rw.hi - rw.lo + THREADS / G - 1) / (THREADS / G);
      float cnt = 0.f;
      if (fuse_loss) {                           // the batch's masked-node count: its loads fly during lin1
        if ((reinterpret_cast<uintptr_t>(a.mask) & 15) == 0) {
          const int nv = a.N >> 4;
          for (int i = tid; i < nv; i += THREADS) {
            const uint4 v = reinterpret_cast<const uint4*>(a.mask)[i];
            auto nz = [](unsigned w) { return __popc((((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u); };
            cnt += (float)(nz(v.x) + nz(v.y) + nz(v.z) + nz(v.w));
          }
          for (int i = (nv << 4) + tid; i < a.N; i += THREADS) cnt += a.mask[i] ? 1.f : 0.f;
        } else {
          for (int i = tid; i < a.N; i += THREADS) cnt += a.mask[i] ? 1.f : 0.f;
        }
      }
      for (int it = 0; it < rounds; ++it) {
        int r = rw.lo + it * (THREADS / G) + tid / G;
        const bool valid = r < rw.hi;
        if (!valid) r = rw.hi - 1;
        const int node = ext_id(a.perm, n0 + r);
        float yv = 0.f;
        bool mk = false;
        if (fuse_loss && valid && (tid % G) == 0) { yv = a.y[node]; mk = a.mask[node] != 0; }
        const float4 xv = ld4(xA + (unsigned)(r * (NC + XPF) + (tid % G) * 4));
        float d = xv.x * wv.x;
        d = fmaf(xv.y, wv.y, d); d = fmaf(xv.z, wv.z, d); d = fmaf(xv.w, wv.w, d);
        for (int off = G >> 1; off > 0; off >>= 1) d += __shfl_xor(d, off);
        if (valid && (tid % G) == 0) {
          const float o = d + bias;
          a.out[node] = o;
          if (fuse_loss) dml[r - rw.lo] = mk ? o - yv : 0.f;
        }
      }
      if (fuse_loss) {
        {                                        // lin1 backward's x operand, before the forward's LDS image dies
          const int c = tid % NC, rg = tid / NC;
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int r = rw.lo + rg + k * LB_R;
            if (r < rw.hi) xk[k] = xA[(unsigned)(r * (NC + XPF) + c)];
          }
        }
        const float Mn = block_sum<THREADS>(cnt, ldsf);            // (its barriers also publish dml)
        const float dv = tid < ow ? dml[tid] : 0.f;
        float part_sum = block_sum<THREADS>(tid < ow ? fmaf(dv, dv, 0.f) : 0.f, ldsf);
        if (tid == 0) {
          a.loss_part[seg * M + part] = part_sum;
          if (seg == 0 && part == 0) a.loss_part[a.num_segments * M] = Mn;
        }
        const float scale = Mn > 0.f ? 2.f / Mn : 0.f;
        if (tid < ow) {
          const float g = dv * scale;
          a.g_out[ext_id(a.perm, n0 + rw.lo + tid)] = g;
          dml[tid] = g;
        }
      }
    }
    __syncthreads();
    STAMP();
    if (fuse_loss) STAMP();    // (diagnostic stamp sequence lin0 | lin1 | loss | lin1_bwd | blocks: the loss is inside lin1 here)
  }

  if ((ph_ & PH_LOSS) && !fuse_loss) {
    float cnt = 0.f;
    for (int i = tid; i < a.N; i += THREADS) cnt += a.mask[i] ? 1.f : 0.f;
    const float Mn = block_sum<THREADS>(cnt, ldsf);
    float part_sum = 0.f;
    for (int r = rw.lo + tid; r < rw.hi; r += THREADS) {
      const int node = ext_id(a.perm, n0 + r);
      if (a.mask[node]) {
        const float d = a.out[node] - a.y[node];
        part_sum = fmaf(d, d, part_sum);
      }
    }
    part_sum = block_sum<THREADS>(part_sum, ldsf);
    if (tid == 0) {
      a.loss_part[seg * M + part] = part_sum;
      if (seg == 0 && part == 0) a.loss_part[a.num_segments * M] = Mn;
    }
    const float scale = Mn > 0.f ? 2.f / Mn : 0.f;
    for (int r = rw.lo + tid; r < rw.hi; r += THREADS) {
      const int node = ext_id(a.perm, n0 + r);
      a.g_out[node] = a.mask[node] ? (a.out[node] - a.y[node]) * scale : 0.f;
    }
    __syncthreads();
    STAMP();
  }

  if (ph_ & GATRES_PHASE_BACKWARD) {
    float* const red = ldsf;
    constexpr bool xedge = NC == 32;                 // win_bwd_dst stores the exported g_e granules itself
    const bool xrows = xedge && keep_;          // ... and so do the dX stages (win_proj) with their rows
    constexpr bool pub = true;           // (the window kernel only runs split segments)
    constexpr int64_t w = 2LL * NC * NC;
    float* gp_cur = sc + L.sc_gpa;
    float* gp_nxt = sc + L.sc_gpb;
    float* const slab = pub ? a.part_slabs + ((int64_t)seg * M + part) * L.slab_stride
                            : a.slabs + (int64_t)seg * L.slab_stride;
    {  // ---- prologue: LDS carve-up, the part's tables, lin1 backward; the geometry goes into the record
    // LDS: red | RA wr x 2NC | ge weg x 2 | gad own x 2 | hT wr x 2NC | asT wr x 2 | adT own x 2 | alT weg x 2 |
    //      xG own x 2NC | W slot A | W slot B | topology | halo lists
    float* RAw = red + 3 * THREADS;
    float* gew = RAw + (size_t)wr * 2 * NC;
    float* gado = gew + 2 * (size_t)even(weg);
    float* hTw = gado + (size_t)((ow * 2 + 3) & ~3);
    float* asTw = hTw + (size_t)wr * 2 * NC;
    float* adTo = asTw + (size_t)((wr * 2 + 3) & ~3);
    float* alTw = adTo + (size_t)((ow * 2 + 3) & ~3);
    float* xGo = alTw + 2 * (size_t)even(weg);
    float* gko = xGo + (size_t)ow * (2 * NC + XPB);                              // keep_: g_pre of the own rows (dX1's residual term)
    float* wlA = reinterpret_cast<float*>(lds_raw + ((reinterpret_cast<unsigned char*>(gko + (keep_ ? (size_t)ow * NC : 0)) - lds_raw + 15) & ~15));
    float* wlB = wlA + WLB;
    u16* tp = reinterpret_cast<u16*>(wlB + WLB);
    u16* rpo = tp;             tp += even(ow + 1);
    u16* colo = tp;            tp += even(oeg);
    u16* trpo = tp;            tp += even(ow + 1);
    u16* teido = tp;           tp += even(otg);
    u16* tdsto = tp;           tp += even(otg);
    u16* mrpw = tp;            tp += even(wr + 1);
    u16* mtrpo = tp;           tp += even(ow + 1);
    u16* mtdsto = tp;          tp += even(otm);
    tp = align16(tp);
    u16* nbin = tp;            tp += 8 * ow;       // padded edge descriptors of the own rows (k_window_stages.h)
    u16* tout = tp;            tp += 16 * ow;
    u16* mout = tp;            tp += 16 * ow;
    int* hcounter = reinterpret_cast<int*>(tp);
    const u16* b_img_end = tp;
    u16* hrow = tp + 2;                  // import lists: remote destinations (+ edge ids) of own out-edges
    unsigned char* bflag_o = lds_top - ((ow + 15) & ~15);      // export flags of the own rows (XOut), from erow
    const int hcap = max(0, (int)((bflag_o - reinterpret_cast<unsigned char*>(hrow)) / 8)) & ~1;
    u16* hedge = hrow + hcap;
    u16* erow = hedge + hcap;            // export lists: own rows with an in-edge from a partner's row, and those in-edges
    u16* eedge = erow + hcap;
    __syncthreads();           // forward's LDS contents are dead from here
    const unsigned char* bflag = bflag_o - lo;
    for (int k = tid; k < ((ow + 15) & ~15) / 4; k += THREADS) reinterpret_cast<unsigned*>(bflag_o)[k] = 0u;
    int hrcnt = 0, hecnt = 0, ercnt = 0, eecnt = 0;      // import rows / import edges / export rows / export edges
    if (PTAB || pt) {
      hrcnt = H(GATRES_PT_B_HRCNT); hecnt = H(GATRES_PT_B_HECNT); ercnt = H(GATRES_PT_B_ERCNT); eecnt = H(GATRES_PT_B_EECNT);
      if (H(GATRES_PT_B_IMG_WORDS) * 2 != (int)(b_img_end - rpo) || hrcnt > hcap || hecnt > hcap || ercnt > hcap || eecnt > hcap) {
        if (tid == 0) *a.err = 1;
        hrcnt = min(hrcnt, hcap); hecnt = min(hecnt, hcap); ercnt = min(ercnt, hcap); eecnt = min(eecnt, hcap);
      } else {
        dma_copy4<THREADS>(reinterpret_cast<float*>(rpo), reinterpret_cast<const float*>(pt + H(GATRES_PT_B_IMG)),
                           H(GATRES_PT_B_IMG_WORDS), 0);
      }
      dma_copy4<THREADS>(reinterpret_cast<float*>(hrow), reinterpret_cast<const float*>(pt + H(GATRES_PT_B_HROW)), (hrcnt + 1) / 2, 0);
      dma_copy4<THREADS>(reinterpret_cast<float*>(hedge), reinterpret_cast<const float*>(pt + H(GATRES_PT_B_HEDGE)), (hecnt + 1) / 2, 0);
      dma_copy4<THREADS>(reinterpret_cast<float*>(erow), reinterpret_cast<const float*>(pt + H(GATRES_PT_B_EROW)), (ercnt + 1) / 2, 0);
      dma_copy4<THREADS>(reinterpret_cast<float*>(eedge), reinterpret_cast<const float*>(pt + H(GATRES_PT_B_EEDGE)), (eecnt + 1) / 2, 0);
    } else {
      copy_rowptr16<THREADS>(rpo, a.rowptr, n0 + lo, ow, e0 + elo);
      copy_idx16<THREADS>(colo, a.col, e0 + elo, oeg, n0);
      copy_rowptr16<THREADS>(trpo, a.t_rowptr, n0 + lo, ow, t0 + tlo);
      copy_idx16<THREADS>(teido, a.t_eid, t0 + tlo, otg, e0);
      copy_idx16<THREADS>(tdsto, a.t_dst, t0 + tlo, otg, n0);
      copy_rowptr16<THREADS>(mrpw, a.m_rowptr, n0 + wlo, wr, a.m_rowptr[n0 + wlo]);
      copy_rowptr16<THREADS>(mtrpo, a.mt_rowptr, n0 + lo, ow, mt0 + mtlo);
      copy_idx16<THREADS>(mtdsto, a.mt_dst, mt0 + mtlo, otm, n0);
    }
    const u16* rp = rpo - lo;  const u16* trp = trpo - lo;  const u16* mrp = mrpw - wlo;  const u16* mtrp = mtrpo - lo;
    float* RA = RAw - wlo * 2 * NC;              // [row][2NC] view: g_out1
    float* gpT = RAw - wlo * NC;                 // [row][NC] views of the lower / upper half: g_pre, g_y2
    float* gy2T = RAw + (size_t)wr * NC - wlo * NC;
    float* ge2 = gew - ewlo;      float* ge1 = gew - ewlo * 2;          // by absolute local edge id
    float* gad2 = gado - lo;      float* gad1 = gado - lo * 2;
    float* hT2 = hTw - wlo * NC;  float* hT1 = hTw - wlo * 2 * NC;
    float* asT2 = asTw - wlo;     float* asT1 = asTw - wlo * 2;
    float* adT2 = adTo - lo;      float* adT1 = adTo - lo * 2;
    float* alT2 = alTw - ewlo;    float* alT1 = alTw - ewlo * 2;
    float* xG2 = xGo - lo * (NC + XPB);   float* xG1 = xGo - lo * (2 * NC + XPB);
    float* gkeep = keep_ ? gko - lo * NC : nullptr;

    const float* xfinal = segbase + (int64_t)L.nb * SL.bstride + SL.xin;
    // LDS-DMA of the saved tables (independent of the backward chain) rides on the dX stages, issued by their tile-less
    // waves, which wait for it to land before the stage's closing barrier.  Every CU streams at the same moments, so a
    // transfer costs its bytes over the CU's share of HBM (~25 GB/s: 1.3 us for conv1's 32 KB).  Measured in round 3
    // and NOT kept: issuing it one or two stages earlier (in the hand-off: the sweeps queue behind the streams, +1.5 us
    // per hand-off; in the source-major stage: that stage grows by what dX shrinks), and dedicated loader waves behind
    // bare barriers (the issue itself is what takes the time: the CU accepts ~12 B per cycle).
    auto dma_conv2_early = [&](int blk, int w0) {
      if (wave_u < w0) return;                   // (the tile waves of the stage: none of the address chains below)
      FRESH_ARGS();
      blk = launder_s(blk);
      const float* bs = segbase + (int64_t)blk * SL.bstride;
      dma_copy16<THREADS>(hTw, bs + SL.h2 + (size_t)wlo * NC, wr * NC, w0);
      dma_copy4<THREADS>(asTw, bs + SL.as2 + wlo, wr, w0);
      dma_copy4<THREADS>(adTo, bs + SL.ad2 + lo, ow, w0);
      w_prefetch<NC, 2 * NC, EPI_RESID_MASK, THREADS>(wlA, a.wt + (int64_t)blk * 2 * w + w, nullptr, nullptr, w0);
      vec_prefetch<THREADS>(wlA + A2OFF, P + L.p_block0 + (int64_t)blk * L.p_block_stride + L.c2_as, 2 * NC);
    };
    auto dma_conv2_late = [&](int blk, int w0) {
      if (wave_u < w0) return;
      FRESH_ARGS();
      blk = launder_s(blk);
      const float* bs = segbase + (int64_t)blk * SL.bstride;
      dma_copy4<THREADS>(alTw, bs + SL.al2 + ewlo, weg, w0);
    };
    if (L.nb > 0) { dma_conv2_early(L.nb - 1, 0); dma_conv2_late(L.nb - 1, 0); }
    if (fuse_loss) {                             // seg_lin1_bwd with g_out from LDS (dml) and x from registers (xk)
      const int c = tid % NC, rg = tid / NC;
      const float wv = P[L.p_lin1_w + c];
      float aw = 0.f, ab = 0.f;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int r = rw.lo + rg + k * LB_R;
        if (r < rw.hi) {
          const float go = dml[r - rw.lo], xv = xk[k];
          aw = fmaf(go, xv, aw);
          ab += go;
          const float gv = (L.nb > 0 && !(xv > 0.f)) ? 0.f : go * wv;
          gp_cur[((size_t)n0 + r) * NC + c] = gv;
          gpT[(size_t)r * NC + c] = gv;
          if (gkeep) gkeep[(size_t)r * NC + c] = gv;
        }
      }
      __syncthreads();
      red[tid] = aw; red[THREADS + tid] = ab;
      __syncthreads();
      if (rg == 0) {
        float s0 = 0.f, s1 = 0.f;
        for (int k = 0; k < LB_R; ++k) { s0 += red[k * NC + c]; s1 += red[THREADS + k * NC + c]; }
        slab[L.p_lin1_w + c] = s0;
        if (c == 0) slab[L.p_lin1_b] = s1;
      }
    } else {
      seg_lin1_bwd<NC, THREADS>(rw, n0, a.perm, a.g_out, xfinal, P + L.p_lin1_w, gp_cur, gpT, slab + L.p_lin1_w,
                                slab + L.p_lin1_b, L.nb > 0 ? 1 : 0, red, gkeep);
    }
    dma_land(0);
    if (PTAB || pt) {
      __syncthreads();         // the record's tables and the first block's conv2 tables have landed
    } else {
      build_nbr_in<THREADS>(rw, rp, colo, oeg, false, nbin);
      build_nbr_out<THREADS>(rw, trp, tdsto, teido, 0, nullptr, otg, tout);
      build_nbr_out<THREADS>(rw, mtrp, mtdsto, nullptr, 0, mrp, otm, mout);
      hrcnt = hecnt = uni(build_halo<THREADS>(trp, tdsto, teido, rw, hrow, hedge, hcap, hcounter));
      ercnt = uni(build_export_rows<THREADS>(a.rowptr, a.col, n0, rw, erow, hcap, hcounter));
      if (tid == 0) *hcounter = 0;
      __syncthreads();
      for (int k = tid; k < oeg; k += THREADS) {               // own in-edges whose source is a partner's row
        const int j = colo[k];
        if (j < lo || j >= rw.hi) {
          const int pos = atomicAdd(hcounter, 1);
          if (pos < hcap) eedge[pos] = (u16)(elo + k);
        }
      }
      __syncthreads();
      eecnt = uni(*hcounter);
      if (hrcnt > hcap || ercnt > hcap || eecnt > hcap) {
        if (tid == 0) *a.err = 1;
        hrcnt = hecnt = min(hrcnt, hcap); ercnt = min(ercnt, hcap); eecnt = min(eecnt, hcap);
      }
    }
    for (int k = tid; k < ercnt; k += THREADS) bflag_o[(int)erow[k] - lo] = 1;
    if constexpr (NC == 32) scale_g_pre<NC, THREADS>(gpT, mrp, lo, ow);      // (lin1 backward's rows; the tables have landed)
    if (tid == 0) {
      BwdRec r;
      r.lo = rw.lo; r.hi = rw.hi; r.ow = ow; r.wr = wr; r.wlo = wlo; r.elo = elo; r.ewlo = ewlo; r.weg = weg; r.n0 = n0;
      r.hrcnt = hrcnt; r.hecnt = hecnt; r.ercnt = ercnt; r.eecnt = eecnt;
      r.RA = lds_off(RA); r.gpT = lds_off(gpT); r.gy2T = lds_off(gy2T); r.ge2 = lds_off(ge2); r.ge1 = lds_off(ge1);
      r.gad2 = lds_off(gad2); r.gad1 = lds_off(gad1); r.hTw = lds_off(hTw); r.hT2 = lds_off(hT2); r.hT1 = lds_off(hT1);
      r.asTw = lds_off(asTw); r.asT2 = lds_off(asT2); r.asT1 = lds_off(asT1); r.adTo = lds_off(adTo); r.adT2 = lds_off(adT2);
      r.adT1 = lds_off(adT1); r.alTw = lds_off(alTw); r.alT2 = lds_off(alT2); r.alT1 = lds_off(alT1); r.xG2 = lds_off(xG2);
      r.xG1 = lds_off(xG1); r.gkeep = gkeep ? lds_off(gkeep) : NO_TABLE; r.wlA = lds_off(wlA); r.wlB = lds_off(wlB);
      r.rp = lds_off(rp); r.colo = lds_off(colo); r.trp = lds_off(trp); r.teido = lds_off(teido); r.tdsto = lds_off(tdsto);
      r.mrp = lds_off(mrp); r.mtrp = lds_off(mtrp); r.mtdsto = lds_off(mtdsto); r.nbin = lds_off(nbin); r.tout = lds_off(tout);
      r.mout = lds_off(mout); r.hrow = lds_off(hrow); r.hedge = lds_off(hedge); r.erow = lds_off(erow); r.eedge = lds_off(eedge);
      r.bflag = lds_off(bflag); r.mo1 = mo1 ? lds_off(mo1) : NO_TABLE; r.mxin = mxin ? lds_off(mxin) : NO_TABLE;
      const int* src = reinterpret_cast<const int*>(&r);
#pragma unroll
      for (int k = 0; k < (int)(sizeof(BwdRec) / 4); ++k)
        __hip_atomic_store(myrec + GATRES_UREC_WORDS / 2 + k, src[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (as in the forward prologue: the barrier does not drain stores)
    }
    }  // ---- (end of the prologue's scope: the block loop below sees the record only)
    __syncthreads();           // export flags and the record are published
    rec_published();
#undef RECP
#define RECP rb_
    auto xout = [&](bool on, const unsigned char* bflag, unsigned t_rows, unsigned t_small) {      // values of the hand-off that follows
      XOut o;
      o.xb = xbuf; o.ep = xc.ep + 1u; o.local = xc.local; o.on = on; o.flag = bflag; o.t_rows = t_rows; o.t_small = t_small;
      return o;
    };
    // LDS-DMA of the saved tables inside the loop (the prologue above issued the first block's with its own pointers)
    auto dma_conv2_early = [&](int blk, int w0) {
      if (wave_u < w0) return;                   // (the tile waves of the stage: none of the address chains below)
      FRESH_ARGS(); FRESH_BWD();
#define LIST_(T_, N_, I_) I_(lo) I_(hi) T_(float, hTw) T_(float, asTw) T_(float, adTo) T_(float, wlA) I_(wlo) I_(wr) I_(ow)
      REC_LOADS(LIST_);
      REC_PINS(LIST_);
      REC_DEFS(LIST_);
#undef LIST_
      blk = launder_s(blk);
      const float* bs = segbase + (int64_t)blk * SL.bstride;
      dma_copy16<THREADS>(hTw, bs + SL.h2 + (size_t)wlo * NC, wr * NC, w0);
      dma_copy4<THREADS>(asTw, bs + SL.as2 + wlo, wr, w0);
      dma_copy4<THREADS>(adTo, bs + SL.ad2 + lo, ow, w0);
      w_prefetch<NC, 2 * NC, EPI_RESID_MASK, THREADS>(wlA, a.wt + (int64_t)blk * 2 * w + w, nullptr, nullptr, w0);
      vec_prefetch<THREADS>(wlA + A2OFF, P + L.p_block0 + (int64_t)blk * L.p_block_stride + L.c2_as, 2 * NC);
    };
    auto dma_conv2_late = [&](int blk, int w0) {
      if (wave_u < w0) return;
      FRESH_ARGS(); FRESH_BWD();
#define LIST_(T_, N_, I_) I_(lo) I_(hi) T_(float, alTw) I_(ewlo) I_(weg)
      REC_LOADS(LIST_);
      REC_PINS(LIST_);
      REC_DEFS(LIST_);
#undef LIST_
      blk = launder_s(blk);
      const float* bs = segbase + (int64_t)blk * SL.bstride;
      dma_copy4<THREADS>(alTw, bs + SL.al2 + ewlo, weg, w0);
    };
    STAMP();
    const int nb = L.nb;
    for (int b = nb - 1; b >= 0; --b) {
      // (per stage: FRESH_ARGS() / FRESH_BWD() re-read the offsets and tables the stage needs; keep / sb / base are formed
      // where they are used)
      [[maybe_unused]] const bool xs_on = STAMPS_PTR && a.stamp_cap >= 4096 && seg == 0 && b == nb / 2;
      [[maybe_unused]] int xs_i = 0;
      XSTAMP();
      // (no barrier in front of exchange B1's sweep when nothing exports here: the sweep writes halo rows only, which the dX1
      // stage still running on slower waves does not touch -- a wave sweeps its share as soon as it arrives, and the LDS-DMA
      // waves' landing wait overlaps with it; the same in front of B3's sweep.  Measured: 0.3728 - 0.3746 -> 0.3691 - 0.3695
      // ms/step; the three forward hand-offs the same way gained nothing and keep theirs.)
      if (NC != 32 || !xrows || b == nb - 1)             // (an export pass reads rows other waves wrote)
      lds_barrier();                                     // own rows of g_pre are in LDS (lin1 backward / the previous dX1)
      XSTAMP();
      {
        FRESH_ARGS(); FRESH_BWD();
#define LIST_(T_, N_, I_) I_(lo) I_(hi) T_(float, gpT) T_(const u16, erow) T_(const u16, hrow) I_(ercnt) I_(hrcnt)
        REC_LOADS(LIST_);
        REC_PINS(LIST_);
        REC_DEFS(LIST_);
#undef LIST_
        ++xc.ep;                                   // exchange B1: K3 backward gathers g_pre of neighbour rows
        if (!xrows || b == nb - 1)                 // (else the previous block's dX1 stored them)
          xch_export2<NC, 0, THREADS>(xc, xbuf, erow, ercnt, gpT, (unsigned)XL.b1, erow, 0, gpT, 0u);
        XSTAMP();
        xch_import2<NC, 0, THREADS>(xc, xbuf, hrow, hrcnt, (unsigned)XL.b1, gpT, hrow, 0, 0u, gpT);
        XSTAMP();
        xch_after<THREADS>(xc, pace, drain);
        XSTAMP();
        lds_barrier();
        XSTAMP();
      }
      if constexpr (!LEAN) publish_items<THREADS>(a, seg, part, 2 * (nb - 1 - b), grp.local, false);      // the blocks above are kept
      // K3 backward, conv2's edge dots and softmax backward: one stage (win_bwd_dst)
      STAMP();
      {
        FRESH_ARGS(); FRESH_BWD();
#define LIST_(T_, N_, I_) I_(lo) I_(hi) T_(const u16, nbin) T_(const u16, rp) T_(const u16, colo) T_(float, gy2T) T_(float, hT2) T_(float, alT2) T_(float, asT2) T_(float, adT2) T_(float, ge2) T_(float, gad2) T_(const u16, mout) T_(const u16, mrp) T_(const u16, mtrp) T_(const u16, mtdsto) T_(float, gpT) T_(const unsigned char, bflag) I_(elo) I_(ow)
        REC_LOADS(LIST_);
        REC_PINS(LIST_);
        REC_DEFS(LIST_);
#undef LIST_
        // Early import of exchange B2: at eight lanes per row a part of up to 64 rows sits on waves 0 .. 7; the other eight run
        // B2's sweep DURING this stage (its halo rows of g_y2 / g_e are not touched by the stage; the partners store their
        // g_y2 granules at the start of their own stage), so the hand-off behind the closing barrier has no sweep of its own:
        // 0.3923 - 0.3936 -> 0.3871 - 0.3903 ms/step.  The same for F3 (behind conv2's aggregation) gave nothing and for B3
        // (three idle waves behind conv1's destination-major stage) cost 8 us: profiles/r04_valu_probe.txt.
        if (NC == 32 && ow64 && wave_u >= 8) {
          const auto rb2_ = rec_fresh<BwdRec>(myrec + GATRES_UREC_WORDS / 2);
          xch_import2_by<NC, 1, 512, THREADS - 512>(xc, xc.ep + 1u, xbuf, LDS_TABLE(const u16, lds_raw, rb2_->hrow), rb2_->hrcnt,
                                                     (unsigned)XL.b2y, gy2T, LDS_TABLE(const u16, lds_raw, rb2_->hedge),
                                                     rb2_->hecnt, (unsigned)XL.b2e, ge2);
        } else
        win_bwd_dst<true, 1, NC, THREADS, NC == 32, NOHUB>(rw, nbin, rp, colo, gy2T, hT2, alT2 + elo, asT2, adT2, ge2 + elo, gad2, mout, mrp,
                                               mtrp, mtdsto, gpT, xout(xedge, bflag, (unsigned)XL.b2y, (unsigned)XL.b2e), elo);
        lds_barrier();
        XSTAMP();
      }
      {
        FRESH_ARGS(); FRESH_BWD();
#define LIST_(T_, N_, I_) I_(lo) I_(hi) T_(float, gy2T) T_(float, ge2) T_(const u16, erow) T_(const u16, eedge) T_(const u16, hrow) T_(const u16, hedge) I_(ercnt) I_(eecnt) I_(hrcnt) I_(hecnt) I_(ow)
        REC_LOADS(LIST_);
        REC_PINS(LIST_);
        REC_DEFS(LIST_);
#undef LIST_
        const bool lean = NC == 32 && ow64;    // (the sweep ran in the stage before; the bias partials ride on the stage after)
        if (!lean) seg_bias_part<NC, THREADS>(rw, gy2T, 0, red);  // (own rows of g_y2: the sweep below only writes halo rows)
        ++xc.ep;                                   // exchange B2: the source-major stage reads g_y2 / g_e of neighbour rows
        if (!xedge) xch_export2<NC, 1, THREADS>(xc, xbuf, erow, ercnt, gy2T, (unsigned)XL.b2y, eedge, eecnt, ge2, (unsigned)XL.b2e);
        XSTAMP();
        if (NC != 32 || !ow64)                   // (else the stage before ran the sweep on its idle waves)
        xch_import2<NC, 1, THREADS>(xc, xbuf, hrow, hrcnt, (unsigned)XL.b2y, gy2T, hedge, hecnt, (unsigned)XL.b2e, ge2);
        XSTAMP();
        xch_after<THREADS>(xc, pace, drain);
        XSTAMP();
        if (!lean) lds_barrier();                  // (lean: nothing between the two stages touches LDS)
        XSTAMP();
        STAMP();
      }
      // this block's conv1 tables and W1^T stream in while the matrix cores run dX2 (below)
      auto dma_conv1_early = [&]() {
        if (wave_u < dw0) return;
        FRESH_ARGS(); FRESH_BWD();
#define LIST_(T_, N_, I_) I_(lo) I_(hi) T_(float, wlB) T_(float, hTw) T_(float, asTw) T_(float, adTo) T_(float, alTw) I_(wlo) I_(wr) I_(ow) I_(ewlo) I_(weg)
        REC_LOADS(LIST_);
        REC_PINS(LIST_);
        REC_DEFS(LIST_);
#undef LIST_
        const int bl = launder_s(b);
        const float* bsl = segbase + (int64_t)bl * SL.bstride;
        w_prefetch<2 * NC, NC, EPI_RESID_MASK, THREADS>(wlB, a.wt + (int64_t)bl * 2 * w, nullptr, nullptr, dw0);
        vec_prefetch<THREADS>(wlB + A1OFF, P + L.p_block0 + (int64_t)bl * L.p_block_stride + L.c1_as, 4 * NC);
        dma_copy16<THREADS>(hTw, bsl + SL.h1 + (size_t)wlo * 2 * NC, wr * 2 * NC, dw0);
        dma_copy4<THREADS>(asTw, bsl + SL.as1 + wlo * 2, wr * 2, dw0);
        dma_copy4<THREADS>(adTo, bsl + SL.ad1 + lo * 2, ow * 2, dw0);
        dma_copy4<THREADS>(alTw, bsl + SL.al1 + ewlo * 2, weg * 2, dw0);      // conv1's alpha (its table held conv2's until here)
      };
      {
        FRESH_ARGS(); FRESH_BWD();
#define LIST_(T_, N_, I_) I_(lo) I_(hi) T_(const u16, tout) T_(const u16, trp) T_(const u16, teido) T_(const u16, tdsto) T_(float, gy2T) T_(float, alT2) T_(float, ge2) T_(float, gad2) T_(float, wlA) T_(float, xG2) I_(n0) I_(ow)
        REC_LOADS(LIST_);
        [[maybe_unused]] float* sb = slab + L.p_block0 + (int64_t)launder_s(b) * L.p_block_stride;
        [[maybe_unused]] float* keep = sc + L.sc_keep + (int64_t)launder_s(b) * L.keep_stride;
        REC_PINS(LIST_, , "s"(sb), "s"(keep));
        REC_DEFS(LIST_);
#undef LIST_
        const bool lean = NC == 32 && ow64;    // rows on waves 0 .. 7: the other eight form conv2's bias partials meanwhile
        if (!lean) seg_bias_finish<NC, THREADS>(red, sb + L.c2_b);
        if (lean && wave_u >= 8) seg_bias_part_by<NC, THREADS, 512, THREADS - 512>(rw, gy2T, 0, red);
        else
        win_agg_bwd_src<1, NC, THREADS, NOHUB, NC + XPB>(rw, tout, trp, teido, tdsto, 0, gy2T, alT2, ge2, gad2, wlA + A2OFF, wlA + A2OFF + NC,
                                             keep + L.k_gh2, n0, keep + L.k_gas2, keep + L.k_gad2, xG2);
        lds_barrier();                   // g_y2 (RA) and the conv2 tables are dead
        if (lean) seg_bias_finish_by<NC, THREADS, THREADS - 64>(red, sb + L.c2_b);      // (the last wave: `red` rests until dst1)
        XSTAMP();
        STAMP();
      }
#ifndef GATRES_PROBE_NO_BWD_DMA       // (timing probe, WRONG results: the backward's block loop without its LDS-DMA of saved tables)
      dma_conv1_early();
#endif
      if (NC == 32 && keep_) {              // (the ReLU sign masks of the forward phase are in LDS: no global operand)
        if constexpr (NC == 32)
          if (wave_u < PW) {
            FRESH_ARGS(); FRESH_BWD();
#define LIST_(T_, N_, I_) I_(lo) I_(hi) I_(ow) T_(float, xG2) T_(float, wlA) T_(float, RA) T_(const unsigned char, bflag) T_(const unsigned long long, mo1)
            REC_LOADS(LIST_);
            REC_PINS(LIST_);
            REC_DEFS(LIST_);
#undef LIST_
            win_proj<NC, 2 * NC, 1, EPI_RESID_MASK, 1, PW, THREADS, NC + XPB>(rw, xG2, wlA, RA, 0, nullptr, nullptr, nullptr, nullptr,
                                                                nullptr, nullptr, nullptr, mo1 + launder_s(b) * ow, nullptr,
                                                                xout(xrows, bflag, (unsigned)XL.b3o, 0u));
          }
      } else {
        FRESH_ARGS(); FRESH_BWD();
#define LIST_(T_, N_, I_) I_(lo) I_(hi) I_(ow) T_(float, xG2) T_(float, wlA) T_(float, RA) N_(const unsigned long long, mo1)
        REC_LOADS(LIST_);
        REC_PINS(LIST_);
        REC_DEFS(LIST_);
#undef LIST_
        const float* base = segbase + (int64_t)launder_s(b) * SL.bstride;
        const float* wt2 = a.wt + (int64_t)launder_s(b) * 2 * w + w;
        seg_proj<NC, 2 * NC, 1, EPI_RESID_MASK, THREADS, true, true>(
            rw, xG2, 0, wt2, RA, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0,
            (GATRES_DIAG && (a.no_halo & 4)) ? nullptr : base + SL.o1, 0, wlA, nullptr, nullptr, mo1 ? mo1 + b * ow : nullptr,
            nullptr);
      }
#ifndef GATRES_PROBE_NO_BWD_LAND      // (timing probe, WRONG results: the backward's dX stages without their LDS-DMA landing waits)
      dma_land(dw0);                             // the conv1 tables: the destination-major stage is next
#endif
      lds_barrier();                             // (not __syncthreads(): its vmcnt(0) made the tile waves wait for their granule stores)
      XSTAMP();
      STAMP();
      {
        FRESH_ARGS(); FRESH_BWD();
#define LIST_(T_, N_, I_) I_(lo) I_(hi) T_(const u16, nbin) T_(const u16, rp) T_(const u16, colo) T_(float, RA) T_(float, hT1) T_(float, alT1) T_(float, asT1) T_(float, adT1) T_(float, ge1) T_(float, gad1) T_(const unsigned char, bflag) I_(elo)
        REC_LOADS(LIST_);
        REC_PINS(LIST_);
        REC_DEFS(LIST_);
#undef LIST_
        seg_bias_part<2 * NC, THREADS>(rw, RA, 0, red);
        win_bwd_dst<false, 2, NC, THREADS, false, NOHUB>(rw, nbin, rp, colo, RA, hT1, alT1 + elo * 2, asT1, adT1, ge1 + elo * 2, gad1,
                                                nullptr, nullptr, nullptr, nullptr, nullptr,
                                                xout(xedge, bflag, 0u, (unsigned)XL.b3e), elo);
        if (NC != 32 || !xrows) lds_barrier();     // (else B3's sweep starts without one: see the top of the loop)
        XSTAMP();
      }
      {
        FRESH_ARGS(); FRESH_BWD();
#define LIST_(T_, N_, I_) I_(lo) I_(hi) T_(float, RA) T_(float, ge1) T_(const u16, erow) T_(const u16, eedge) T_(const u16, hrow) T_(const u16, hedge) I_(ercnt) I_(eecnt) I_(hrcnt) I_(hecnt)
        REC_LOADS(LIST_);
        REC_PINS(LIST_);
        REC_DEFS(LIST_);
#undef LIST_
        ++xc.ep;                                   // exchange B3
        if (!xrows || !xedge)
          xch_export2<2 * NC, 2, THREADS>(xc, xbuf, erow, xrows ? 0 : ercnt, RA, (unsigned)XL.b3o, eedge, xedge ? 0 : eecnt, ge1,
                                          (unsigned)XL.b3e);
        XSTAMP();
        xch_import2<2 * NC, 2, THREADS>(xc, xbuf, hrow, hrcnt, (unsigned)XL.b3o, RA, hedge, hecnt, (unsigned)XL.b3e, ge1);
        XSTAMP();
        xch_after<THREADS>(xc, pace, drain);
        XSTAMP();
        lds_barrier();
        XSTAMP();
      }
      if constexpr (!LEAN) publish_items<THREADS>(a, seg, part, 2 * (nb - 1 - b) + 1, grp.local, false);      // conv2 tables complete
      STAMP();
      {
        FRESH_ARGS(); FRESH_BWD();
#define LIST_(T_, N_, I_) I_(lo) I_(hi) T_(const u16, tout) T_(const u16, trp) T_(const u16, teido) T_(const u16, tdsto) T_(float, RA) T_(float, alT1) T_(float, ge1) T_(float, gad1) T_(float, wlB) T_(float, xG1) I_(n0)
        REC_LOADS(LIST_);
        [[maybe_unused]] float* sb = slab + L.p_block0 + (int64_t)launder_s(b) * L.p_block_stride;
        [[maybe_unused]] float* keep = sc + L.sc_keep + (int64_t)launder_s(b) * L.keep_stride;
        REC_PINS(LIST_, , "s"(sb), "s"(keep));
        REC_DEFS(LIST_);
#undef LIST_
        seg_bias_finish<2 * NC, THREADS>(red, sb + L.c1_b);
        win_agg_bwd_src<2, NC, THREADS, NOHUB, 2 * NC + XPB>(rw, tout, trp, teido, tdsto, 0, RA, alT1, ge1, gad1, wlB + A1OFF, wlB + A1OFF + 2 * NC,
                                             keep + L.k_gh1, n0, keep + L.k_gas1, keep + L.k_gad1, xG1);
        lds_barrier();
        XSTAMP();
        STAMP();
      }
#ifndef GATRES_PROBE_NO_BWD_DMA
      if (b > 0) { dma_conv2_early(b - 1, dw0); dma_conv2_late(b - 1, dw0); }
#endif
      if (NC == 32 && keep_) {
        if constexpr (NC == 32)
          if (wave_u < PW) {
            FRESH_ARGS(); FRESH_BWD();
#define LIST_(T_, N_, I_) I_(lo) I_(hi) I_(ow) I_(n0) T_(float, xG1) T_(float, wlB) T_(float, gpT) T_(const unsigned char, bflag) T_(float, gkeep) T_(const unsigned, mxin) T_(const u16, mrp)
            REC_LOADS(LIST_);
            REC_PINS(LIST_);
            REC_DEFS(LIST_);
#undef LIST_
            const int bl = launder_s(b);
            win_proj<2 * NC, NC, 1, EPI_RESID_MASK, 1, PW, THREADS, 2 * NC + XPB>(rw, xG1, wlB, gp_nxt, n0, gpT, nullptr, nullptr, nullptr,
                                                                nullptr, gkeep, gkeep, nullptr, bl > 0 ? mxin + bl * ow : nullptr,
                                                                xout(xrows && bl > 0, bflag, (unsigned)XL.b1, 0u), mrp);
          }
      } else {
        FRESH_ARGS(); FRESH_BWD();
#define LIST_(T_, N_, I_) I_(lo) I_(hi) I_(ow) I_(n0) T_(float, xG1) T_(float, wlB) T_(float, gpT) N_(float, gkeep) N_(const unsigned, mxin) T_(const u16, mrp)
        REC_LOADS(LIST_);
        REC_PINS(LIST_);
        REC_DEFS(LIST_);
#undef LIST_
        const float* base = segbase + (int64_t)launder_s(b) * SL.bstride;
        const float* wt1 = a.wt + (int64_t)launder_s(b) * 2 * w;
        seg_proj<2 * NC, NC, 1, EPI_RESID_MASK, THREADS, true, true>(
            rw, xG1, 0, wt1, gp_nxt, n0, gpT, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
            (GATRES_DIAG && (a.no_halo & 4)) ? nullptr : gp_cur, n0,
            (b > 0 && !(GATRES_DIAG && (a.no_halo & 4))) ? base + SL.xin : nullptr, 0, wlB, gkeep, gkeep, nullptr,
            (mxin && b > 0) ? mxin + b * ow : nullptr);
        if constexpr (NC == 32) {                  // (rare path) g_pre in LDS as K3 backward gathers it: see scale_g_pre
          __syncthreads();
          scale_g_pre<NC, THREADS>(gpT, mrp, lo, ow);
        }
      }
#ifndef GATRES_PROBE_NO_BWD_LAND
      dma_land(dw0);
#endif
      XSTAMP();
      STAMP();
      float* t = gp_cur; gp_cur = gp_nxt; gp_nxt = t;
    }
    FRESH_ARGS(); FRESH_BWD();
#define LIST_(T_, N_, I_) I_(lo) I_(hi) N_(float, gkeep) T_(float, gpT) I_(n0)
    REC_LOADS(LIST_);
    REC_PINS(LIST_);
    REC_DEFS(LIST_);
#undef LIST_
    // The parts only meet here for the consumers' sake (the last items are published behind this barrier); without consumer
    // workgroups nothing of a partner is needed any more: a workgroup barrier that also drains this part's own stores (g_x below
    // reads gp_cur back) replaces the flag barrier.  a.C is the same for every part: the barrier count per launch stays equal.
    if (assume_local) group_verify_local(grp);      // (one L2 round trip on wave 0, under the barrier below)
    if (nC_ > 0) group_sync<THREADS>(grp);
    else         __syncthreads();
    if constexpr (!LEAN) publish_items<THREADS>(a, seg, part, 2 * L.nb, grp.local || !pub, !pub);
    // lin0's partial sums: g_pre of the own rows from LDS when the launch keeps them there (gkeep; gpT holds them divided by
    // the in-degrees for NC == 32), else from gp_cur
    seg_lin0_bwd<NC, THREADS>(rw, n0, a.perm, gp_cur, a.x, a.mask, slab + L.p_lin0_w, slab + L.p_lin0_b, red,
                              NC == 32 ? gkeep : gpT,
#ifndef GATRES_NO_XKEEP
                              (NC == 32 && keep_ && L.nb > 0 && (ph_ & GATRES_PHASE_FORWARD)) ? mxin : nullptr
#else
                              nullptr
#endif
                              );
    if (pub && nC_ > 0) {
      group_sync<THREADS>(grp);
      publish_items<THREADS>(a, seg, part, 2 * L.nb + 1, grp.local, true);
    }
    if (a.g_x) {
      constexpr int G = NC / 4;
      const float4 wv = ld4(P + L.p_lin0_w + (tid % G) * 4);
      const int rounds = (rw.hi - rw.lo + THREADS / G - 1) / (THREADS / G);
      for (int it = 0; it < rounds; ++it) {
        int r = rw.lo + it * (THREADS / G) + tid / G;
        const bool valid = r < rw.hi;
        if (!valid) r = rw.hi - 1;
        const float4 xv = ld4(gp_cur + ((size_t)n0 + r) * NC + (tid % G) * 4);
        float d = xv.x * wv.x;
        d = fmaf(xv.y, wv.y, d); d = fmaf(xv.z, wv.z, d); d = fmaf(xv.w, wv.w, d);
        for (int off = G >> 1; off > 0; off >>= 1) d += __shfl_xor(d, off);
        if (valid && (tid % G) == 0) a.g_x[ext_id(a.perm, n0 + r)] = d;
      }
    }
  }
  if (tid == 0) __hip_atomic_store(grp.flags + part * FLAG_STRIDE + 2, xc.ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (M > 1 && tid == 0 && *a.err) {
    if ((ph_ & GATRES_PHASE_FORWARD) && a.out) a.out[n0] = NAN;
    if (ph_ & GATRES_PHASE_BACKWARD) a.slabs[(int64_t)seg * L.slab_stride + L.p_lin1_b] = NAN;
  }
  if (STAMPS_PTR && blockIdx.x == 0 && threadIdx.x == 0) {
    STAMPS_PTR[a.stamp_cap + 1] = clock64();
    STAMPS_PTR[a.stamp_cap + 2] = wall_clock64();
  }
}

}  // namespace

extern "C" __attribute__((visibility("hidden"))) int gatres_fused_launch_window(const void* args, int nc, unsigned grid, void* stream) {
  const FusedArgs& a = *static_cast<const FusedArgs*>(args);
  hipStream_t st = gatres_stream(stream);
  switch (nc) {
    case 4: hipLaunchKernelGGL((gatres_window_kernel<4, 1024>), dim3(grid), dim3(1024), 0, st, a); break;
    case 8: hipLaunchKernelGGL((gatres_window_kernel<8, 1024>), dim3(grid), dim3(1024), 0, st, a); break;
    case 16: hipLaunchKernelGGL((gatres_window_kernel<16, 1024>), dim3(grid), dim3(1024), 0, st, a); break;
    case 32: {
      // the launch's compile-time facts (gatres_window_kernel): phases | 0x100 keep-in-LDS | 0x200 symmetric plan, no consumer
      // workgroups | 0x400 rows of at most MAXD entries | 0x800 part tables | 0x1000 parts of at most 64 rows
      int facts = ((a.sym && a.C == 0) ? 0x200 : 0) | (a.facts & 0x1400) | ((a.ptab && a.ptab_m == a.M) ? 0x800 : 0);
      if ((a.facts & 0x2000) && facts == 0x1e00) facts |= 0x2000;      // (inference: only with every other fact -- one instantiation)
      if ((a.facts & 0x4000) && (facts & 0x1e00) == 0x1e00 && !a.safe_sync) facts |= 0x4000;      // (probed dispatch: likewise)
      const int key = a.phases | (a.keep_lds ? 0x100 : 0) | (facts & gatres_knobs()->window_ph_mask);
      if (!gatres_knobs()->window_runtime_phases) {
        switch (key) {
          case 0x5f16: hipLaunchKernelGGL((gatres_window_kernel<32, 1024, 0x5f16>), dim3(grid), dim3(1024), 0, st, a); return gatres_launch_status();
          case 0x7e02: hipLaunchKernelGGL((gatres_window_kernel<32, 1024, 0x7e02>), dim3(grid), dim3(1024), 0, st, a); return gatres_launch_status();
          case 0x5e02: hipLaunchKernelGGL((gatres_window_kernel<32, 1024, 0x5e02>), dim3(grid), dim3(1024), 0, st, a); return gatres_launch_status();
          case 0x5e04: hipLaunchKernelGGL((gatres_window_kernel<32, 1024, 0x5e04>), dim3(grid), dim3(1024), 0, st, a); return gatres_launch_status();
          case 0x1f16: hipLaunchKernelGGL((gatres_window_kernel<32, 1024, 0x1f16>), dim3(grid), dim3(1024), 0, st, a); return gatres_launch_status();
          case 0x1b16: hipLaunchKernelGGL((gatres_window_kernel<32, 1024, 0x1b16>), dim3(grid), dim3(1024), 0, st, a); return gatres_launch_status();      // (a plan with hub rows)
          case 0x0f16: hipLaunchKernelGGL((gatres_window_kernel<32, 1024, 0x0f16>), dim3(grid), dim3(1024), 0, st, a); return gatres_launch_status();
          case 0x0716: hipLaunchKernelGGL((gatres_window_kernel<32, 1024, 0x0716>), dim3(grid), dim3(1024), 0, st, a); return gatres_launch_status();
          case 0x0316: hipLaunchKernelGGL((gatres_window_kernel<32, 1024, 0x0316>), dim3(grid), dim3(1024), 0, st, a); return gatres_launch_status();
          case 0x3e02: hipLaunchKernelGGL((gatres_window_kernel<32, 1024, 0x3e02>), dim3(grid), dim3(1024), 0, st, a); return gatres_launch_status();
          case 0x1e02: hipLaunchKernelGGL((gatres_window_kernel<32, 1024, 0x1e02>), dim3(grid), dim3(1024), 0, st, a); return gatres_launch_status();
          case 0x1e04: hipLaunchKernelGGL((gatres_window_kernel<32, 1024, 0x1e04>), dim3(grid), dim3(1024), 0, st, a); return gatres_launch_status();
          case 0x116: hipLaunchKernelGGL((gatres_window_kernel<32, 1024, 0x116>), dim3(grid), dim3(1024), 0, st, a); return gatres_launch_status();
          default: break;
        }
      }
      hipLaunchKernelGGL((gatres_window_kernel<32, 1024>), dim3(grid), dim3(1024), 0, st, a);
      break;
    }
    default: return GATRES_E_UNSUPPORTED;
  }
  return gatres_launch_status();
}
