// sampling.hip -- the non-greedy branch of the decode loop on the device:
// reference magma/sampling.py:99-107
//     if top_k > 0: logits = top_k_filter(logits, k=top_k)          (:22-30)
//     if top_p > 0: logits = top_p_filter(logits, threshold=top_p)  (:7-19, the reference's own rule -- SURVEY Q6)
//     probs = softmax(logits / temperature); next_token = multinomial(probs, 1)
// and the early-stop test (next_token == eos).all() (:109), as ONE launch per token step + a one-thread bookkeeping
// launch, both graph-capturable: no host round trip per step (the reference syncs once per token, and runs the filters
// as ~10 PyTorch launches including a full sort of the 50 258 logits).
//
// One workgroup of 1024 threads per row; the row (200 KB of fp32 at V = 50 258) stays in L2 and is swept a few times:
//   top-k        threshold = the k-th largest logit by MSB-first radix selection on order-preserving 32-bit keys
//                (4 sweeps, integer histograms in LDS); every logit below it becomes -inf; of the ties AT the threshold the
//                first ones by index stay, so that exactly k survive (torch.topk keeps an unspecified subset of them --
//                equal values, so the probability mass the next stage sees is the same).
//   "top-p"      the reference sorts descending, accumulates softmax probabilities and DROPS every rank r >= 1 whose
//                preceding mass cum[r-1] is still below 1 - threshold.  No sort is needed for that: an element is dropped
//                iff it is not the first maximum and the probability mass of the strictly larger logits is < 1 - threshold,
//                and that mass is monotone in the logit -- so the set is {x >= t*} minus the maximum, with t* the smallest
//                logit whose strictly-larger mass is below the bound.  t* comes from the same radix descent with MASS
//                histograms, preceded (round 3) by one level of 256 LINEAR value bins over [max(min, max - 32), max]: the
//                key's top byte puts nearly all 50 258 logits into two or three buckets -- one LDS atomic queue -- and a
//                monotone partition of any shape keeps the descent's invariant; the 256 buckets of a level are scanned by
//                one wave (suffix sums by shuffles, the monotone predicate by ballot) instead of one thread.  172 -> ~120 us
//                per step for 8 rows.  Masses are 2^-40 fixed point in 64-bit integers: integer atomics are associative, so the
//                result does not depend on the order in which the lanes arrive (fp32 atomics would make a sampled token
//                depend on scheduling).  Ties at t* are dropped one by one in index order, as a stable sort would.
//   multinomial  inverse-CDF draw in index order over the same fixed-point weights exp((x - max) / temperature):
//                target = floor(r * total / 2^64) with r = 64 Philox4x32-10 bits keyed by (seed, step, row); prefix sums
//                are integer, hence exact and reproducible (tests restate the draw in Python integers / float64).
#include "common.h"

namespace {

constexpr int ST = 1024;

MG_DEV uint32_t fkey(float x) {            // order-preserving map float -> uint32 (larger float, larger key)
  const uint32_t b = __float_as_uint(x);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// ---- deterministic block reductions (fixed tree, 16 waves) ----
MG_DEV float blk_max(float v, float* sh) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sh[0];
#pragma unroll
  for (int w = 1; w < ST / 64; ++w) r = fmaxf(r, sh[w]);
  return r;
}
MG_DEV float blk_sum(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sh[0];
#pragma unroll
  for (int w = 1; w < ST / 64; ++w) r += sh[w];
  return r;
}
MG_DEV int blk_min_i(int v, int* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  int r = sh[0];
#pragma unroll
  for (int w = 1; w < ST / 64; ++w) r = min(r, sh[w]);
  return r;
}

// Philox4x32-10 (Salmon et al. 2011), counter (c0..c3), key (k0, k1)
MG_DEV void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

struct SampleParams {
  const float* logits; int64_t ld; int V;
  float temperature; int top_k; double top_p;
  const uint64_t* seed;       // device: 64-bit seed (may change between graph replays)
  const int32_t* state;       // device: [0] = step counter (read here, advanced by sample_finish_kernel)
  int64_t* out;               // [B] sampled token (nullptr: filter only)
  float* filtered; int64_t ldf;   // optional [B, V] filtered logits (tests / callers that want the reference's tensor)
};

constexpr double FIX = 1099511627776.0;   // 2^40

__global__ __launch_bounds__(ST) void sample_kernel(const SampleParams p) {
  __shared__ unsigned long long hist64[256];
  __shared__ uint32_t hist32[256];
  __shared__ float shf[ST / 64];
  __shared__ int shi[ST / 64];
  __shared__ unsigned long long shu[ST / 64 + 1];
  __shared__ uint32_t bc[4];
  __shared__ unsigned long long bc64[2];
  const int tid = threadIdx.x, V = p.V;
  const float* x = p.logits + (int64_t)blockIdx.x * p.ld;

  // One wave scans the 256 buckets of a top-p level (it used to be one thread walking them: ~8 us per level): the lowest
  // non-empty bucket whose strictly-larger mass -- `start` plus the buckets above it -- is below `limit`.  That mass only grows
  // downwards, so the predicate is monotone: suffix sums by shuffles, the lowest qualifying bucket by ballot.
  auto scan_mass = [&](unsigned long long start, unsigned long long limit) {
    if (tid < 64) {
      unsigned long long m[4]; uint32_t c[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { m[j] = hist64[4 * tid + j]; c[j] = hist32[4 * tid + j]; }
      const unsigned long long lane_sum = m[0] + m[1] + m[2] + m[3];
      unsigned long long incl = lane_sum;                   // sum over lanes >= this one
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long t = __shfl_down(incl, o, 64);
        if (tid + o < 64) incl += t;
      }
      const unsigned long long s3 = start + (incl - lane_sum);    // mass strictly above bucket 4 * tid + 3
      const unsigned long long s2 = s3 + m[3], s1 = s2 + m[2], s0 = s1 + m[1];
      int j = -1; unsigned long long sj = 0;
      if (c[0] && s0 < limit) { j = 0; sj = s0; }
      else if (c[1] && s1 < limit) { j = 1; sj = s1; }
      else if (c[2] && s2 < limit) { j = 2; sj = s2; }
      else if (c[3] && s3 < limit) { j = 3; sj = s3; }
      const unsigned long long has = __ballot(j >= 0);
      if (has == 0ull) { if (tid == 0) { bc[0] = 0u; bc[1] = 0u; bc64[0] = start; bc64[1] = 0ull; } }
      else if (tid == (int)__builtin_ctzll(has)) { bc[0] = (uint32_t)(4 * tid + j); bc[1] = c[j]; bc64[0] = sj; bc64[1] = m[j]; }
    }
  };

  // ---------------- top-k: key of the k-th largest logit ----------------
  uint32_t tk = 0;                                  // alive(i) = key > tk || (key == tk && i < k_cut): exactly k survive
  int k_cut = 0x7fffffff;
  if (p.top_k > 0 && p.top_k < V) {
    uint32_t prefix = 0, mask = 0, remaining = (uint32_t)p.top_k;
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (tid < 256) hist32[tid] = 0;
      __syncthreads();
      for (int i = tid; i < V; i += ST) {
        const uint32_t k = fkey(x[i]);
        if ((k & mask) == prefix) atomicAdd(&hist32[(k >> shift) & 255], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        uint32_t rem = remaining; int d = 0;
        for (int b = 255; b >= 0; --b) {
          if (hist32[b] >= rem) { d = b; break; }
          rem -= hist32[b];
        }
        bc[0] = (uint32_t)d; bc[1] = rem; bc[2] = hist32[d];
      }
      __syncthreads();
      prefix |= bc[0] << shift; mask |= 255u << shift; remaining = bc[1];
      const uint32_t in_bucket = bc[2];
      __syncthreads();
      if (shift == 0 && in_bucket > remaining) {     // more ties at the threshold than places left: the first ones by index stay
        if (tid == 0) {
          uint32_t seen = 0; int cut = 0x7fffffff;
          for (int i = 0; i < V; ++i) if (fkey(x[i]) == prefix) { if (seen == remaining) { cut = i; break; } ++seen; }
          bc[3] = (uint32_t)cut;
        }
        __syncthreads();
        k_cut = (int)bc[3];
        __syncthreads();
      }
    }
    tk = prefix;
  }
  auto alive = [&](int i, uint32_t k) -> bool { return k > tk || (k == tk && i < k_cut); };

  // ---------------- first maximum (rank 0 of the reference's sort; never dropped) ----------------
  float mx = -INFINITY, mn = INFINITY;      // mn: smallest finite logit (only spans the top-p level-0 bins)
  for (int i = tid; i < V; i += ST) { const float v = x[i]; mx = fmaxf(mx, v); if (v > -INFINITY) mn = fminf(mn, v); }
  mx = blk_max(mx, shf);
  mn = -blk_max(-mn, shf);
  int imax = 0x7fffffff;
  for (int i = tid; i < V; i += ST) if (x[i] == mx) { imax = min(imax, i); }
  imax = blk_min_i(imax, shi);

  // ---------------- the reference's top-p rule ----------------
  uint32_t tstar = 0xffffffffu;    // dropped(i) = alive && i != imax && (key > tstar || (key == tstar && i < tie_cut))
  int tie_cut = 0;
  // the reference compares an fp32 tensor with the PYTHON scalar (1 - threshold): the scalar is evaluated in double, then cast
  // to fp32 for the comparison -- top_p arrives as a double so that exactly that value is formed (0.1f for 0.9, not 0.10000002f)
  const double bound = (double)(float)(1.0 - p.top_p);
  if (p.top_p > 0.0 && bound > 0.0) {
    float z = 0.f;
    for (int i = tid; i < V; i += ST) if (alive(i, fkey(x[i]))) z += __expf(x[i] - mx);
    z = blk_sum(z, shf);
    const float rz = 1.0f / z;
    const unsigned long long cfix = (unsigned long long)(bound * FIX);
    // Level 0: 256 LINEAR value bins over [max(min, max - 32), max] (below max - 32 the fixed-point mass is 0), then the four 8-bit levels of
    // the order-preserving key inside the chosen bin.  Any monotone partition keeps the descent's invariant (`above` = mass
    // strictly above the current range), so t* is what four key levels alone would find -- but the key's top byte (sign + high
    // exponent bits) puts nearly all of the 50 000 logits into two or three buckets, i.e. one LDS atomic queue; linear bins
    // spread them.
    const float lo = fmaxf(mn, mx - 32.0f);
    const float bscale = mx > lo ? 255.99f / (mx - lo) : 0.f;
    auto bin0 = [&](float v) -> int { const int b = (int)((v - lo) * bscale); return v > lo ? min(b, 255) : 0; };
    uint32_t prefix = 0, mask = 0;
    unsigned long long above = 0;     // mass of the alive keys strictly above the current range
    uint32_t n_eq = 0;
    int b0 = 0;
    for (int level = 0; level < 5; ++level) {
      const int shift = 24 - 8 * (level - 1);            // levels 1..4: key bytes, most significant first
      if (tid < 256) { hist64[tid] = 0; hist32[tid] = 0; }
      __syncthreads();
      for (int i = tid; i < V; i += ST) {
        const float v = x[i];
        const uint32_t k = fkey(v);
        if (alive(i, k)) {
          const int vb = bin0(v);
          if (level == 0 || (vb == b0 && (k & mask) == prefix)) {
            const int bucket = level == 0 ? vb : (int)((k >> shift) & 255);
            const unsigned long long q = (unsigned long long)((double)(__expf(v - mx) * rz) * FIX + 0.5);
            atomicAdd(&hist64[bucket], q);
            atomicAdd(&hist32[bucket], 1u);
          }
        }
      }
      __syncthreads();
      scan_mass(above, cfix);
      __syncthreads();
      if (level == 0) b0 = (int)bc[0];
      else { prefix |= bc[0] << shift; mask |= 255u << shift; n_eq = bc[1]; }
      above = bc64[0];
      __syncthreads();
    }
    tstar = prefix;
    // ties at t*: the j-th of them (index order) has preceding mass above + j * q_eq
    int m = (int)n_eq;
    if (n_eq > 1) {
      const unsigned long long q_eq = bc64[1] / n_eq;
      unsigned long long s = above; m = 0;
      while (m < (int)n_eq && s < cfix) { ++m; s += q_eq; }
    }
    tie_cut = 0x7fffffff;
    if (m < (int)n_eq) {           // rare: only the first m ties (by index) go -- one thread finds the (m+1)-th tie's index
      if (tid == 0) {
        int seen = 0, cut = 0x7fffffff;
        for (int i = 0; i < V; ++i) if (fkey(x[i]) == tstar && alive(i, tstar)) { if (seen == m) { cut = i; break; } ++seen; }
        bc[2] = (uint32_t)cut;
      }
      __syncthreads();
      tie_cut = (int)bc[2];
    }
  }
  auto kept = [&](int i, float v) -> bool {
    const uint32_t k = fkey(v);
    if (!alive(i, k)) return false;
    if (i == imax) return true;
    return !(k > tstar || (k == tstar && i < tie_cut));
  };

  if (p.filtered) {
    float* f = p.filtered + (int64_t)blockIdx.x * p.ldf;
    for (int i = tid; i < V; i += ST) { const float v = x[i]; f[i] = kept(i, v) ? v : -INFINITY; }
  }
  if (!p.out) return;

  // ---------------- multinomial draw: inverse CDF over fixed-point weights, index order ----------------
  // Wave w owns the contiguous index range [w * RW * 64, (w + 1) * RW * 64) as RW rows of 64: lane l reads index
  // seg0 + j * 64 + l, so every load is one coalesced 256-byte row (a contiguous range PER THREAD made each load instruction
  // touch 64 cache lines).  Index order inside the range is (row, lane): the wave that holds the target walks its rows with a
  // lane scan per row.  Same integers, same order of the CDF: the same token as before.
  const int lane = tid & 63, wave = tid >> 6;
  const int RW = (V + ST - 1) / ST;
  const int seg0 = wave * RW * 64;
  const float rt = 1.0f / p.temperature;
  auto weight = [&](int i) -> unsigned long long {
    if (i >= V) return 0ull;
    const float v = x[i];
    return kept(i, v) ? (unsigned long long)((double)__expf((v - mx) * rt) * 4294967296.0 + 0.5) : 0ull;
  };
  unsigned long long wsum = 0;
  for (int j = 0; j < RW; ++j) wsum += weight(seg0 + j * 64 + lane);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wsum += __shfl_xor(wsum, o, 64);          // total of this wave's range, in every lane
  __syncthreads();
  if (lane == 0) shu[wave] = wsum;
  __syncthreads();
  if (tid == 0) {
    unsigned long long run = 0;
    for (int w = 0; w < ST / 64; ++w) { const unsigned long long t = shu[w]; shu[w] = run; run += t; }
    shu[ST / 64] = run;
  }
  __syncthreads();
  const unsigned long long excl = shu[wave], total = shu[ST / 64];
  uint32_t c[4] = {(uint32_t)p.state[0], blockIdx.x, 0u, 0u};
  const unsigned long long seed = p.seed ? *p.seed : 0ull;
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const unsigned long long r = ((unsigned long long)c[0] << 32) | c[1];
  const unsigned long long target = __umul64hi(r, total);           // uniform in [0, total)
  if (wsum > 0 && target >= excl && target < excl + wsum) {         // exactly one wave (the condition is wave-uniform)
    unsigned long long run = excl;
    for (int j = 0; j < RW; ++j) {
      const int i = seg0 + j * 64 + lane;
      unsigned long long incl = weight(i);
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
      }
      const unsigned long long row_total = __shfl(incl, 63, 64);
      if (run + row_total > target) {                               // the token is in this row
        const unsigned long long hit = __ballot(run + incl > target);
        if (lane == 0) p.out[blockIdx.x] = seg0 + j * 64 + (int)__builtin_ctzll(hit);     // first index whose inclusive CDF exceeds the target
        break;
      }
      run += row_total;
    }
  }
  if (total == 0 && tid == 0) p.out[blockIdx.x] = imax == 0x7fffffff ? 0 : imax;   // nothing representable: the maximum
}

// Loop bookkeeping of one token step in ONE small launch: (next_token == eos).all() of reference sampling.py:109 recorded as
// the FIRST step at which it held, the step counter of the random stream, the KV-cache write position, and the token
// history the host reads once at the end of generate() (instead of one small copy per step).
// state = {step, first_all_eos_step (-1 = not yet)}
__global__ void sample_finish_kernel(const int64_t* __restrict__ tok, int B, int64_t eos, int32_t* __restrict__ state,
                                     int32_t* __restrict__ d_pos, int delta, int64_t* __restrict__ history, int64_t ld_hist,
                                     int hist_cols, int32_t* __restrict__ clear, int n_clear, int clear_stride) {
  // every thread needs the step BEFORE thread 0 bumps it: one read, broadcast through LDS (rows beyond the first wave would
  // otherwise race with the increment below)
  __shared__ int s_step;
  if (threadIdx.x == 0) s_step = state[0];
  __syncthreads();
  const int step = s_step;
  for (int i = threadIdx.x; i < n_clear; i += blockDim.x) clear[(int64_t)i * clear_stride] = 0;
  if (history && step < hist_cols)
    for (int b = threadIdx.x; b < B; b += blockDim.x) history[(int64_t)b * ld_hist + step] = tok[b];
  if (threadIdx.x != 0) return;
  bool all = true;
  for (int b = 0; b < B; ++b) all = all && (tok[b] == eos);
  if (all && state[1] < 0) state[1] = step;
  state[0] = step + 1;
  if (d_pos) *d_pos += delta;
}

}  // namespace

extern "C" int mg_sample_f32(const float* logits, int64_t ld, int32_t B, int32_t V, float temperature, int32_t top_k,
                             double top_p, const uint64_t* seed, const int32_t* state, int64_t* token, float* filtered,
                             int64_t ld_filtered, void* stream) {
  if (B <= 0 || V <= 0 || !logits || ld < V) MG_FAIL(MG_ERR_SHAPE, "mg_sample_f32: bad logits / B / V / ld");
  if (!token && !filtered) MG_FAIL(MG_ERR_SHAPE, "mg_sample_f32: nothing to produce (token and filtered are both null)");
  if (token && (!state || !(temperature > 0.f))) MG_FAIL(MG_ERR_SHAPE, "mg_sample_f32: sampling needs temperature > 0 and a state buffer (greedy decoding is mg_argmax_f32)");
  if (top_k < 0 || top_p < 0.0 || top_p > 1.0) MG_FAIL(MG_ERR_SHAPE, "mg_sample_f32: need top_k >= 0 and 0 <= top_p <= 1");
  if (filtered && ld_filtered < V) MG_FAIL(MG_ERR_SHAPE, "mg_sample_f32: ld_filtered < V");
  SampleParams p{logits, ld, V, temperature, top_k, top_p, seed, state, token, filtered, ld_filtered};
  hipLaunchKernelGGL(sample_kernel, dim3(B), dim3(ST), 0, (hipStream_t)stream, p);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_sample_finish(const int64_t* token, int32_t B, int64_t eos, int32_t* state, int32_t* d_pos, int32_t delta,
                                int64_t* history, int64_t ld_history, int32_t history_cols, int32_t* clear, int32_t n_clear,
                                int32_t clear_stride, void* stream) {
  if (!token || !state || B <= 0) MG_FAIL(MG_ERR_SHAPE, "mg_sample_finish: bad arguments");
  if (history && (ld_history < history_cols || history_cols <= 0)) MG_FAIL(MG_ERR_SHAPE, "mg_sample_finish: bad history geometry");
  hipLaunchKernelGGL(sample_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, token, B, eos, state, d_pos, delta,
                     history, ld_history, history_cols, clear, clear ? n_clear : 0, clear_stride > 0 ? clear_stride : 1);
  MG_CHECK_LAUNCH();
  return MG_OK;
}
