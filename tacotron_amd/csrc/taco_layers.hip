// Once-per-utterance stages for gfx950: input vectors, Conv1d + BatchNorm + activation tiles
// (encoder and postnet), the encoder BiLSTM recurrence, speaker concat + attention memory projection.
#include "taco_device.h"

#include <type_traits>

namespace {

// ------------------------------------------------------------------------------------------------
// Input vectors: concatenated per-channel embedding rows / one-hot blocks, channel-major [B][E][T].
// reference: get_vector + cat + transpose, src/tacotron/model.py:752-773, 585-609.
// ------------------------------------------------------------------------------------------------
struct EmbedArgs {
  const int64_t* ids[TACO_MAX_CHANNELS];
  const float* table[TACO_MAX_CHANNELS];
  int width[TACO_MAX_CHANNELS];
  int rows[TACO_MAX_CHANNELS];
  int n_channels, E, T, B;
  const int32_t* lens;
  float* x;
  uint32_t* bad_id;      // set to 1 + channel when an id lies outside its table
};

__global__ __launch_bounds__(256) void taco_embed_kernel(const EmbedArgs a) {
  const size_t total = (size_t)a.B * a.E * a.T;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int t = (int)(i % a.T);
    const int e = (int)((i / a.T) % a.E);
    const int b = (int)(i / ((size_t)a.T * a.E));
    const int len = a.lens ? a.lens[b] : a.T;
    float v = 0.0f;
    if (t < len) {
      int off = 0;
#pragma unroll
      for (int c = 0; c < TACO_MAX_CHANNELS; ++c) {
        if (c < a.n_channels && e >= off && e < off + a.width[c]) {
          const int64_t id = a.ids[c][(size_t)b * a.T + t];
          if (id >= 0 && id < a.rows[c]) {
            if (a.table[c]) v = a.table[c][(size_t)id * a.width[c] + (e - off)];
            else v = (id == (int64_t)(e - off)) ? 1.0f : 0.0f;
          } else {
            *a.bad_id = 1u + (uint32_t)c;
          }
        }
        if (c < a.n_channels) off += a.width[c];
      }
    }
    a.x[i] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// Conv1d(kernel KS, zero padding (KS-1)/2) fused with a per-channel affine (eval BatchNorm folded to
// y*alpha+beta), an activation and an optional residual add, on the f32-input matrix cores
// (v_mfma_f32_32x32x2_f32: exact f32, each product rounded once, same as an fmaf chain).
//
// The convolution is the GEMM  out[co][t] = sum_{ci,kk} w[co][ci][kk] * in[ci][t + kk - PAD].
// One 256-thread workgroup computes a 32-channel x 64-column output tile.  Its four waves are
// (column half) x (half of the input channels of a slab): every wave owns one 32 x 32 accumulator
// tile and the two channel halves are added through LDS at the end.  Input columns and weights are
// staged through LDS in slabs of CI input channels, double-buffered, with the global loads running
// TWO slabs ahead in registers: one slab's MFMAs (~1 us) do not cover a load's round trip.
// MFMA operand maps (32x32x2): lane l supplies A[row l&31][k = l>>5] and B[k = l>>5][col l&31]; the
// accumulator holds column l&31, rows (r&3) + 8*(r>>2) + 4*(l>>5) in register r.
// reference: ConvNorm (src/tacotron/layers.py:17-35) + BatchNorm1d in Encoder.inference
// (src/tacotron/model.py:280-289) and Postnet.forward (205-211); residual add model.py:788-789.
// ------------------------------------------------------------------------------------------------
constexpr int CV_CO = 32, CV_TT = 64, CV_XS = 96;
constexpr int taco_conv_ci(int ks) { return ks == 1 ? 64 : 32; }
constexpr size_t taco_conv_lds_bytes(int ks) {
  return sizeof(float) * 2 * ((size_t)taco_conv_ci(ks) * CV_XS + (size_t)CV_CO * (taco_conv_ci(ks) * ks + 2));
}
typedef float taco_f32x16 __attribute__((ext_vector_type(16)));

template <int KS, bool IN_TM>
__global__ __launch_bounds__(256) void taco_conv_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        const float* __restrict__ alpha, const float* __restrict__ beta,
                                                        const float* __restrict__ residual, const int32_t* lens,
                                                        int c_in, int c_out, int T, int act, int layout,
                                                        int split, float* split_ws, int split_cap) {
  constexpr bool in_time_major = IN_TM;          // input is [T][C] instead of [C][T]
  const bool out_time_major = layout & 2;        // output likewise
  constexpr int CI = taco_conv_ci(KS);         // input channels per slab
  constexpr int PAD = (KS - 1) / 2;
  constexpr int XW = CV_TT + KS - 1;           // input columns a tile needs
  constexpr int XS = CV_XS;                    // (a B-operand read: the 32 lanes of a half wave take 32 consecutive words - conflict-free)
  constexpr int WS = CI * KS + 2;              // even row stride, so that a thread's staged weights go to LDS as 8-byte writes.  A
                                               // ds_read_b32 is banked by word address mod 32 within each half wave, so the 32 rows
                                               // of a k-lane meet two to a bank (one extra LDS cycle per A-operand read); an odd
                                               // stride (conflict-free reads and writes, built and measured) needs twice the write
                                               // instructions and came out 4 % slower: the slab loop waits for issue slots, not
                                               // for the LDS array
  constexpr int RL = CI * KS;                  // weights of one output channel in a slab: contiguous in memory
  constexpr int NW = RL / 8;                   // 8 threads per weight row, NW consecutive floats each
  constexpr int HW = XW - CV_TT;               // halo columns of an input row beyond the first 64 (KS - 1)
  constexpr bool ROWMAP = !IN_TM;              // input staging map, see below
  constexpr int NX = (XW + 7) / 8, XR = CI / 32; // (position-major inputs: 8 threads per input row, columns seg, seg + 8, ...)
  constexpr int NXI = ROWMAP ? CI / 4 + (HW > 0 ? 1 : 0) : XR * NX;   // input words per thread and slab
  static_assert(XW <= XS && 2 * CV_CO * WS >= 2 * 16 * 64 && RL % 32 == 0 && CI * 8 == 256 * (CI / 32) && NW % 2 == 0 && (2 * CI * XS) % 2 == 0 && CI * HW <= 256, "tile geometry");
  extern __shared__ __attribute__((aligned(16))) float conv_lds[];
  float (*xs)[CI][XS] = reinterpret_cast<float (*)[CI][XS]>(conv_lds);                       // [2][CI][XS]
  float (*ws)[CV_CO][WS] = reinterpret_cast<float (*)[CV_CO][WS]>(conv_lds + 2 * CI * XS);   // [2][CV_CO][WS]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int thalf = wave & 1, khalf = wave >> 1;
  const int i = lane & 31, h = lane >> 5;
  const int b = blockIdx.z;
  // split > 1: `split` consecutive workgroups share an output tile, each with one unit of slabs (see the end of the
  // slab loop for how their partial tiles meet)
  // layout & 4: the grid is (channel blocks, column tiles, utterances) instead of (column tiles x split, channel blocks,
  // utterances).  Workgroups go to the eight XCDs round-robin in launch order, so what runs fastest through the grid decides
  // what an XCD gets: with the column tiles there and 16 tiles of which a ragged batch fills only the first few, XCD 0 (tiles
  // 0 and 8) has twice the work of XCD 7 (tiles 7 and 15, mostly past the utterances' ends).  Channel blocks are always
  // all there: every XCD gets the same share of every utterance, and an XCD's L2 keeps the weights of its own blocks.
  const bool by_block = layout & 4;
  const int tile_t = by_block ? blockIdx.y : (split > 1 ? blockIdx.x / split : blockIdx.x);
  const int ksl = (!by_block && split > 1) ? blockIdx.x % split : 0;
  const int t0 = tile_t * CV_TT;
  const int co0 = (by_block ? blockIdx.x : blockIdx.y) * CV_CO;
  const int len = lens ? min(lens[b], T) : T;      // (a length beyond the buffers' T columns is read as T)
  const float* in_b = in + (size_t)b * c_in * T;
  // acc: a unit's accumulation chain (and, after the slab loop, the tile's sum); acc1: the chain of every other unit in
  // the loop's steady state; tot: the sum of the units before the running one
  taco_f32x16 acc, acc1, tot;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = acc1[r] = tot[r] = 0.0f;

  if (t0 < len) {
    // Staging map.  Weights: thread = (row tid >> 3, segment tid & 7), row = output channel, the thread takes NW
    // consecutive floats of the slab's RL (16-byte loads).  Inputs, channel-major: wave w takes rows w * CI / 4 + j,
    // lane = column (the 64 lanes of a load or an LDS write cover 64 consecutive words: whole cache lines, 32 distinct
    // banks per half wave), and the HW halo columns of all CI rows are one more word for thread (row tid / HW, column
    // 64 + tid % HW).  Inputs, position-major (memory runs along the channels): thread (row tid >> 3, segment tid & 7)
    // takes CI / 32 rows (row + 32 q), columns seg + 8 n.
    // Both come through buffer loads, whose out-of-range offsets return zero: a padding position, a column
    // past the utterance, an input channel past c_in or a weight row past c_out is simply given the offset ~0 (no
    // branch, no 64-bit address arithmetic, nothing read that is not there).  A ragged last slab's weight loads run on
    // into the next row's weights - finite values that meet input rows staged as zeros.
    constexpr int LW = NW / 4, LX = NXI;         // load items of a slab per thread: 16-byte weight loads, input words
    constexpr int SW = NW / 2, SX = NXI;         // store items: 8-byte weight writes, input words
    constexpr int ITEMS = LW + LX + SW + SX;
    constexpr int MFMAS = KS * (CI / 4);         // matrix-core products of a slab per wave
    constexpr unsigned OOB = 0xffffffffu;
    typedef unsigned taco_u4 __attribute__((ext_vector_type(4)));
    const int srow = tid >> 3, seg = tid & 7;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)((unsigned)c_out * (unsigned)c_in * KS * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)in_b, 0, (int)((unsigned)c_in * (unsigned)T * 4u), 0x00020000);
    const unsigned w_off = co0 + srow < c_out ? ((unsigned)(co0 + srow) * (unsigned)c_in * KS + (unsigned)(seg * NW)) * 4u : OOB;
    // byte offset of my column inside a row (channel-major) / of its row (position-major): main part, halo part
    const int hrow = HW > 0 ? tid / (HW > 0 ? HW : 1) : CI, hcol = HW > 0 ? CV_TT + tid % (HW > 0 ? HW : 1) : 0;
    unsigned col_off[ROWMAP ? 2 : NX];
#pragma unroll
    for (int n = 0; n < (ROWMAP ? 2 : NX); ++n) {
      const int c = ROWMAP ? (n == 0 ? lane : hcol) : seg + 8 * n, tt = t0 + c - PAD;
      const bool mine = ROWMAP ? (n == 0 || hrow < CI) : c < XW;
      col_off[n] = (mine && tt >= 0 && tt < len) ? (unsigned)tt * (in_time_major ? (unsigned)c_in : 1u) * 4u : OOB;
    }
    float pwA[NW], pxA[NXI], pwB[NW], pxB[NXI];
    // item `it` of the loads of the slab that starts at input channel ci0
    auto load_item = [&](int it, int ci0, float (&pw)[NW], float (&px)[NXI]) {
      if (it < LW) {
        const unsigned off = w_off == OOB ? OOB : w_off + (unsigned)(ci0 * KS + 4 * it) * 4u;
        const taco_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0);
        pw[4 * it] = __uint_as_float(v.x); pw[4 * it + 1] = __uint_as_float(v.y);
        pw[4 * it + 2] = __uint_as_float(v.z); pw[4 * it + 3] = __uint_as_float(v.w);
      } else {
        const int j = it - LW;
        if constexpr (ROWMAP) {
          const bool halo = j >= CI / 4;
          const int ci = ci0 + (halo ? hrow : wave * (CI / 4) + j);
          const unsigned coff = col_off[halo ? 1 : 0];
          const unsigned off = (coff != OOB && ci < c_in) ? coff + (unsigned)ci * (unsigned)T * 4u : OOB;
          px[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_x, off, 0, 0));
        } else {
          const int q = j / NX, n = j % NX;
          const int ci = ci0 + srow + 32 * q;
          const unsigned off = (col_off[n] != OOB && ci < c_in) ? col_off[n] + (unsigned)ci * 4u : OOB;
          px[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_x, off, 0, 0));
        }
      }
    };
    auto store_item = [&](int it, int buf, const float (&pw)[NW], const float (&px)[NXI]) {
      if (it < SW) {
        *reinterpret_cast<float2*>(&ws[buf][srow][seg * NW + 2 * it]) = make_float2(pw[2 * it], pw[2 * it + 1]);
      } else {
        const int j = it - SW;
        if constexpr (ROWMAP) {
          if (j < CI / 4) xs[buf][wave * (CI / 4) + j][lane] = px[j];
          else if (hrow < CI) xs[buf][hrow][hcol] = px[j];
        } else {
          const int q = j / NX, n = j % NX;
          if (seg + 8 * n < XW) xs[buf][srow + 32 * q][seg + 8 * n] = px[j];
        }
      }
    };
    const int n_slabs_all = (c_in + CI - 1) / CI;
    // my slabs: sl0 .. sl0 + n_slabs - 1 (at least one) - all of them, or unit ksl when every unit has its own workgroup
    const int sl0 = split > 1 ? ksl * TACO_CONV_UNIT_SLABS : 0;
    const int n_slabs = split > 1 ? min(TACO_CONV_UNIT_SLABS, n_slabs_all - sl0) : n_slabs_all;
    // One slab: multiply slab s (LDS buffer s & 1), fetch slab s + 2 into `fetch`, park slab s + 1 (`park`, fetched one
    // slab ago) in the other LDS buffer.  A wave's products form one dependent chain (a product issues when the one
    // before has left the matrix core, 64 cycles), and a wave issues in order: whatever stands between two products
    // runs in the first one's shadow, whatever is bunched up behind the last one does not.  So the slab's staging work
    // - the operand reads of the next step, the global loads, the LDS writes - is dealt out one or two instructions at
    // a time behind each product, and scheduling barriers keep the compiler from gathering it up again.
    // `cur` is the running unit's chain.  FOLD (a unit's first slab in the steady state): the unit before ran on `prev`; its
    // sixteen registers join `tot` one at a time behind this slab's first sixteen products - done right at the unit's end
    // they would stop the product chain for the length of a slab's fortieth (the adds wait for the last product, the next
    // product for the cleared registers).
    auto slab = [&](auto fold_c, taco_f32x16& cur, taco_f32x16& prev, int s, float (&fw)[NW], float (&fx)[NXI], const float (&kw)[NW],
                    const float (&kx)[NXI]) {
      constexpr bool FOLD = decltype(fold_c)::value;
      const int buf = s & 1;
      const int ci_next = (sl0 + min(s + 2, n_slabs - 1)) * CI;     // past the end: a harmless reload of the last slab
      float av[2][KS], bv[2][KS];
      auto read_pair = [&](int step, int set, int kk) {
        const int cl = khalf * (CI / 2) + step * 2 + h;    // this lane's input channel of the k-pair
        av[set][kk] = ws[buf][i][cl * KS + kk];
        bv[set][kk] = xs[buf][cl][thalf * 32 + i + kk];
      };
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) read_pair(0, 0, kk);
#pragma unroll
      for (int step = 0; step < CI / 4; ++step) {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
          const int m = step * KS + kk;
          if (step + 1 < CI / 4) read_pair(step + 1, (step + 1) & 1, kk);
          cur = __builtin_amdgcn_mfma_f32_32x32x2f32(av[step & 1][kk], bv[step & 1][kk], cur, 0, 0, 0);
          if constexpr (FOLD) {
            if (m < 16) { tot[m] += prev[m]; prev[m] = 0.0f; }
          }
#pragma unroll
          for (int it = m * ITEMS / MFMAS; it < (m + 1) * ITEMS / MFMAS; ++it) {
            if (it < LW + LX) load_item(it, ci_next, fw, fx);
            else store_item(it - (LW + LX), buf ^ 1, kw, kx);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
    };
#pragma unroll
    for (int it = 0; it < LW + LX; ++it) load_item(it, sl0 * CI, pwA, pxA);
#pragma unroll
    for (int it = 0; it < LW + LX; ++it) load_item(it, (sl0 + min(1, n_slabs - 1)) * CI, pwB, pxB);
#pragma unroll
    for (int it = 0; it < SW + SX; ++it) store_item(it, 0, pwA, pxA);
    __syncthreads();
    // (Both slabs of a pair unconditionally inside the loop, the odd one behind it: with `if (s + 1 < n_slabs)` around
    // the second, the compiler sees a path from the first slab straight back to the loop's top on which the first
    // slab's loads are still in flight, and drains every load at the top of each pair.)
    static_assert(TACO_CONV_UNIT_SLABS == 4 && MFMAS >= 16, "the steady state below spells out two units of four slabs");
    const std::true_type fold;
    const std::false_type plain;
    int s = 0;
    // steady state: two units per round, on alternating chains; a unit's sum joins `tot` during the next unit's first slab
    // (in unit order: the additions are the ones of the plain loop below)
    for (; s + 7 < n_slabs; s += 8) {
      slab(fold, acc, acc1, s, pwA, pxA, pwB, pxB);
      slab(plain, acc, acc1, s + 1, pwB, pxB, pwA, pxA);
      slab(plain, acc, acc1, s + 2, pwA, pxA, pwB, pxB);
      slab(plain, acc, acc1, s + 3, pwB, pxB, pwA, pxA);
      slab(fold, acc1, acc, s + 4, pwA, pxA, pwB, pxB);
      slab(plain, acc1, acc, s + 5, pwB, pxB, pwA, pxA);
      slab(plain, acc1, acc, s + 6, pwA, pxA, pwB, pxB);
      slab(plain, acc1, acc, s + 7, pwB, pxB, pwA, pxA);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) tot[r] += acc1[r];             // the steady state's last unit (zeros if it never ran)
    for (; s + 1 < n_slabs; s += 2) {
      slab(plain, acc, acc1, s, pwA, pxA, pwB, pxB);
      slab(plain, acc, acc1, s + 1, pwB, pxB, pwA, pxA);
      if (((s + 2) & (TACO_CONV_UNIT_SLABS - 1)) == 0) {      // a unit is complete: its chain ends here
#pragma unroll
        for (int r = 0; r < 16; ++r) { tot[r] += acc[r]; acc[r] = 0.0f; }
      }
    }
    if (s < n_slabs) slab(plain, acc, acc1, s, pwA, pxA, pwB, pxB);
    // (a last unit of fewer slabs; after a whole number of units this adds zeros)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = tot[r] + acc[r];
    if (split > 1) {
      // Every workgroup of the tile leaves its unit's four accumulator tiles in global memory (write-through 16-byte
      // stores, so no release fence is needed) and draws a ticket from the tile's counter; the one that draws the last
      // ticket adds all units from memory - its own included, in unit order, every wave its own tile: the very sums a
      // single workgroup running all units would have made - and goes on.  It also puts the counter back to zero for the
      // next launch that uses it.  Partial of unit k of tile n: floats [(n * split + k) * 4096 ..) as
      // [wave][register quad][lane][4].
      const unsigned tiles_t = gridDim.x / (unsigned)split;
      const unsigned tile = ((unsigned)b * gridDim.y + blockIdx.y) * tiles_t + (unsigned)tile_t;
      const unsigned n_part = gridDim.z * gridDim.y * tiles_t * (unsigned)split;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)split_ws, 0, (int)(n_part * (TACO_CONV_PARTIAL * 4u)), 0x00020000);
      unsigned* counters = reinterpret_cast<unsigned*>(split_ws + (size_t)split_cap * TACO_CONV_PARTIAL);
      const unsigned lane_off = ((unsigned)wave * 4u * 64u + (unsigned)lane) * 16u;
      {
        const unsigned at = (tile * (unsigned)split + (unsigned)ksl) * (TACO_CONV_PARTIAL * 4u) + lane_off;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const taco_u4 v = {__float_as_uint(acc[4 * q]), __float_as_uint(acc[4 * q + 1]), __float_as_uint(acc[4 * q + 2]),
                             __float_as_uint(acc[4 * q + 3])};
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, at + (unsigned)q * 1024u, 0, 16 /* sc1: write-through */);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave's stores have been acknowledged
      __syncthreads();
      unsigned* ticket_lds = reinterpret_cast<unsigned*>(conv_lds);      // the input buffers are no longer in use
      if (tid == 0) *ticket_lds = __hip_atomic_fetch_add(counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      if (*ticket_lds != (unsigned)split - 1u) return;
      if (tid == 0) __hip_atomic_store(counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
      for (int k = 0; k < split; ++k) {
        const unsigned at = (tile * (unsigned)split + (unsigned)k) * (TACO_CONV_PARTIAL * 4u) + lane_off;
        taco_u4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, at + (unsigned)q * 1024u, 0, 16 /* sc1: past this CU's L1 */);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[4 * q] += __uint_as_float(v[q].x);
          acc[4 * q + 1] += __uint_as_float(v[q].y);
          acc[4 * q + 2] += __uint_as_float(v[q].z);
          acc[4 * q + 3] += __uint_as_float(v[q].w);
        }
      }
      __syncthreads();                                      // (the ticket word lies in the buffers used next)
    }
    // add the two channel halves: waves 2, 3 hand their tiles to waves 0, 1 through LDS
    float* red = &ws[0][0][0];     // 2 x 16 x 64 floats fit the weight buffers
    if (khalf == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(thalf * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (khalf == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += red[(thalf * 16 + r) * 64 + lane];
    }
  }
  if (khalf != 0) return;
  const int t = t0 + thalf * 32 + i;
  if (t >= T) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (co >= c_out) continue;
    float v = 0.0f;
    if (t < len) {
      const float al = alpha ? alpha[co] : 1.0f;
      const float be = beta ? beta[co] : 0.0f;
      const float bi = bias ? bias[co] : 0.0f;
      v = (acc[r] + bi) * al + be;
      if (act == 1) v = v > 0.0f ? v : 0.0f;
      else if (act == 2) v = tanhf(v);
      if (residual) v += residual[((size_t)b * c_out + co) * T + t];
    }
    if (out_time_major) out[((size_t)b * T + t) * c_out + co] = v;
    else out[((size_t)b * c_out + co) * T + t] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// Encoder BiLSTM recurrence, as many utterances per launch as fit the device at once (each with its own
// workgroups and hand-off words; utterances never talk to each other).  The input projections W_ih x_t + b_ih + b_hh
// of all steps are computed beforehand (a kernel-size-1 convolution); this kernel runs the sequential
// part.  Few, fat workgroups: a hand-off between 9 workgroups is much cheaper than one between 65.
// A workgroup of 1024 threads owns 32 hidden units of one direction = 128 gate rows, 8 lanes per row:
// lane `part` of a row keeps the row's columns 32k + 4 part .. +3 in registers (one ds_read_b128 of the
// hidden vector per k), the 8 partial sums meet by DPP, the 4 gates of a unit sit in 32 adjacent lanes
// of one wave.  The hidden state travels through tagged granules (double-buffered by step parity,
// EL_REP replicas).
// reference: nn.LSTM(bidirectional) in Encoder.inference, src/tacotron/model.py:253-260, 280-289;
// for padded batches the reverse direction starts at each utterance's own last symbol
// (pack_padded_sequence semantics, model.py:262-278).
// ------------------------------------------------------------------------------------------------
constexpr int EL_UNITS = 32;          // hidden units per workgroup
constexpr int EL_HMAX = 512;          // hidden width capacity
constexpr int EL_REP = 1;             // replicas of the hidden-state vector (1, 2, 4, 8 measured: 340, 343, 352, 388 us per encode)
constexpr int EL_GRAN_DIR = 2 * EL_REP * EL_HMAX;   // granules per direction: 2 parities x replicas x width

struct EncLstmArgs {
  const float* w_hh[2];   // [4H][H]
  const float* gx;        // [utterances][2][4H][T], first utterance of this launch
  float* memory;          // [utterances][T][M], likewise
  const int32_t* lens;    // device, or NULL = T symbols
  int b;                  // index of this launch's first utterance into lens
  int H, T, M, n_wg_dir;
  taco_granule* gran;     // [utterances of a launch][2 dirs][2 parities][EL_REP][EL_HMAX]
  uint32_t* ctl;
};

// KC = 32-column chunks of a row held in registers (4 weights per lane and chunk): H <= 32 KC
template <int KC>
__global__ __launch_bounds__(1024) void taco_enc_lstm_kernel(const EncLstmArgs a) {
  __shared__ __attribute__((aligned(16))) float xh[EL_HMAX];
  __shared__ int gave_up;
  const int tid = threadIdx.x, lane = tid & 63;
  const int utt = blockIdx.y;                      // utterance of this launch; blockIdx.x = direction, slice of units
  const int wg = blockIdx.x;
  const int len = a.lens ? min(a.lens[a.b + utt], a.T) : a.T;
  const int dir = wg / a.n_wg_dir;
  const int jj = wg % a.n_wg_dir;
  const int r = tid >> 3, part = tid & 7;          // gate row of this workgroup, 8 lanes per row
  const int unit = r >> 2, gate_id = r & 3;        // rows 4u..4u+3 = gates i, f, g, o of unit u
  const int gu = jj * EL_UNITS + unit;
  const bool active = gu < a.H;
  const int row = gate_id * a.H + gu;
  float wr[KC][4];
#pragma unroll
  for (int k = 0; k < KC; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int col = 32 * k + 4 * part + e;
      wr[k][e] = (active && col < a.H) ? a.w_hh[dir][(size_t)row * a.H + col] : 0.0f;
    }
  for (int i = tid; i < EL_HMAX; i += 1024) xh[i] = 0.0f;
  if (tid == 0) gave_up = 0;
  __syncthreads();

  taco_wait_ctx wctx;
  wctx.abort_word = (taco_gu32*)a.ctl;
  wctx.failed = 0;
  wctx.spun = 0;
  taco_granule* gdir = a.gran + ((size_t)utt * 2 + dir) * EL_GRAN_DIR;
  const float* gx = a.gx + ((size_t)utt * 2 + dir) * 4 * a.H * a.T;
  float* memory = a.memory + (size_t)utt * a.T * a.M;
  const bool gate_lane = active && part == 0;      // finishes its row's gate
  const bool unit_lane = gate_lane && gate_id == 0;  // finishes its unit's cell update
  float cell = 0.0f;
  const int my_rep = wg % EL_REP;

  for (int s = 0; s < len; ++s) {
    const int t = dir == 0 ? s : len - 1 - s;
    float gx_v = 0.0f;
    if (gate_lane) gx_v = gx[(size_t)row * a.T + t];   // in flight during the wait below
    if (s > 0) {
      // collect h_{s-1}; a give-up is passed on through LDS, so that the barrier orders LDS traffic only
      // (a full __syncthreads would also wait for this wave's publish stores to be acknowledged: ~1 us)
      if (tid < a.H) xh[tid] = taco_wait_one(gdir + ((size_t)((s - 1) & 1) * EL_REP + my_rep) * EL_HMAX, tid, (unsigned)s, wctx, 0xE00u);
      if (wctx.failed) gave_up = 1;
      taco_lds_barrier();
      if (gave_up) break;
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // every chunk of the hidden vector is asked for before the first product: the reads' latency (not their
    // number) is what this loop would otherwise wait for
    float4 xv[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) xv[k] = *reinterpret_cast<const float4*>(xh + 32 * k + 4 * part);
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      acc[0] = fmaf(wr[k][0], xv[k].x, acc[0]);
      acc[1] = fmaf(wr[k][1], xv[k].y, acc[1]);
      acc[2] = fmaf(wr[k][2], xv[k].z, acc[2]);
      acc[3] = fmaf(wr[k][3], xv[k].w, acc[3]);
    }
    const float pre = gx_v + taco_sum8((acc[0] + acc[1]) + (acc[2] + acc[3]));
    const float act = gate_id == 2 ? taco_tanh(pre) : taco_sigmoid(pre);     // valid in the row's lane 0
    // the unit's four gates sit 8 lanes apart in this half-wave
    const int base = lane & 32;
    const float ig = __shfl(act, base + 0, 64);
    const float fg = __shfl(act, base + 8, 64);
    const float gg = __shfl(act, base + 16, 64);
    const float og = __shfl(act, base + 24, 64);
    if (unit_lane) {
      cell = fg * cell + ig * gg;
      const float hv = og * taco_tanh(cell);
      taco_publish_rep<EL_REP>(gdir + (size_t)(s & 1) * EL_REP * EL_HMAX, EL_HMAX, gu, (unsigned)s + 1u, hv);
      memory[(size_t)t * a.M + dir * a.H + gu] = hv;
    }
    taco_lds_barrier();  // every thread is done with xh before the next step overwrites it (LDS only: the
                         // publish stores stay in flight)
  }
}

// ------------------------------------------------------------------------------------------------
// Speaker vector concat (one workgroup per (b, t)); processed_memory = memory_layer(memory) follows as
// a kernel-size-1 convolution.
// reference: src/tacotron/model.py:779-782 (speaker concat), 403 (memory_layer, no bias).
// ------------------------------------------------------------------------------------------------
struct TailArgs {
  const int64_t* speakers;
  const float* spk_table;
  const int32_t* lens;
  float* memory;           // [B][T][M]
  int T, M, S, H2, spk_rows;
  uint32_t* bad_id;
};

__global__ __launch_bounds__(256) void taco_memory_tail_kernel(const TailArgs a) {
  const int t = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x;
  const int len = a.lens ? a.lens[b] : a.T;
  float* mem = a.memory + ((size_t)b * a.T + t) * a.M;
  if (t >= len) {
    for (int k = tid; k < a.M; k += 256) mem[k] = 0.0f;
    return;
  }
  if (a.S > 0) {
    const int64_t id = a.speakers[(size_t)b * a.T + t];
    for (int k = tid; k < a.S; k += 256) {
      float v = 0.0f;
      if (id >= 0 && id < a.spk_rows) v = a.spk_table ? a.spk_table[(size_t)id * a.S + k] : (id == k ? 1.0f : 0.0f);
      else *a.bad_id = 1u + (uint32_t)TACO_MAX_CHANNELS;
      mem[a.H2 + k] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Strided column-slice copy, used once per handle to stack the context columns of several matrices.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void taco_slice_cols_kernel(const float* __restrict__ src, int ld, int col0,
                                                              int rows, int cols, float* __restrict__ dst) {
  const size_t total = (size_t)rows * cols;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t r = i / cols;
    const int c = (int)(i % cols);
    dst[i] = src[r * ld + col0 + c];
  }
}

// Location layers folded: out[a][c][k] = sum_f dense[a][f] * conv[f][c][k]  (rows a >= A are zero).
// reference: LocationLayer = Conv1d(2 -> F, K, no bias) then Linear(F -> A, no bias), src/tacotron/model.py:20-44.
__global__ __launch_bounds__(256) void taco_loc_fold_kernel(const float* __restrict__ dense_w, const float* __restrict__ conv_w,
                                                            int A, int F, int K, int rows, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * 2 * K) return;
  const int a = i / (2 * K), ck = i % (2 * K);
  float acc = 0.0f;
  if (a < A)
    for (int f = 0; f < F; ++f) acc = fmaf(dense_w[(size_t)a * F + f], conv_w[(size_t)f * 2 * K + ck], acc);
  out[i] = acc;
}

}  // namespace

int taco_launch_loc_fold(const float* dense_w, const float* conv_w, int A, int F, int K, int rows, float* out, hipStream_t s) {
  const int n = rows * 2 * K;
  hipLaunchKernelGGL(taco_loc_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dense_w, conv_w, A, F, K, rows, out);
  TACO_HIP(hipGetLastError());
  return TACO_OK;
}

int taco_launch_slice_cols(const float* src, int ld, int col0, int rows, int cols, float* dst, hipStream_t s) {
  const size_t total = (size_t)rows * cols;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(taco_slice_cols_kernel, dim3(blocks), dim3(256), 0, s, src, ld, col0, rows, cols, dst);
  TACO_HIP(hipGetLastError());
  return TACO_OK;
}

int taco_launch_embed(taco_handle h, const int64_t* const* ids_dev, const int32_t* lens, int B, int T,
                      float* x, hipStream_t s) {
  EmbedArgs a;
  const taco_dims& d = h->d;
  for (int c = 0; c < TACO_MAX_CHANNELS; ++c) {
    a.ids[c] = c < d.n_channels ? ids_dev[c] : nullptr;
    a.table[c] = c < d.n_channels ? h->w.ch_table[c] : nullptr;
    a.width[c] = c < d.n_channels ? d.ch_width[c] : 0;
    a.rows[c] = c < d.n_channels ? d.ch_rows[c] : 0;
    if (c < d.n_channels && !a.ids[c]) TACO_FAIL(TACO_EINVAL, "ids[%d] is NULL", c);
  }
  a.n_channels = d.n_channels; a.E = d.E; a.T = T; a.B = B; a.lens = lens; a.x = x; a.bad_id = h->enc_ctl + 2;
  const size_t total = (size_t)B * d.E * T;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(taco_embed_kernel, dim3(blocks), dim3(256), 0, s, a);
  TACO_HIP(hipGetLastError());
  return TACO_OK;
}

int taco_conv_prepare() {
  // the slab buffers exceed the default 64 KiB of dynamic LDS for the wider kernels
#define TACO_CONV_ATTR(KS)                                                                                      \
  TACO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(taco_conv_kernel<KS, false>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                               (int)taco_conv_lds_bytes(KS)));                                                                  \
  TACO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(taco_conv_kernel<KS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                               (int)taco_conv_lds_bytes(KS)));
  TACO_CONV_ATTR(1)
  TACO_CONV_ATTR(3)
  TACO_CONV_ATTR(5)
  TACO_CONV_ATTR(7)
  TACO_CONV_ATTR(9)
#undef TACO_CONV_ATTR
  return TACO_OK;
}

// How many workgroups share an output tile.  A convolution with fewer tiles than the device has CUs runs at the latency
// of one workgroup's serial slab loop, so the loop is cut into up to 8 ranges of at least two slabs each, as long as the
// launch stays within one workgroup per CU (and within the partial-tile workspace).  Measured on the encoder
// (102 symbols: 34 tiles -> 7 workgroups each, 37 -> 20 us per layer) and the postnet (250 frames: 171 -> 95 us for
// the five layers; 1000 frames: the 80-channel last layer 41 -> 23 us); filling the device twice or cutting the
// 256-tile layers in two gained nothing: two workgroups on a CU take as long as one with twice the slabs.
static int taco_conv_split_factor(int base_tiles, int n_slabs, int n_cu, int split_cap) {
  // all or nothing: every unit of slabs gets a workgroup of its own, or one workgroup runs them all (either way the
  // units' sums are added in the same order, so the choice never shows in the result)
  const int units = (n_slabs + TACO_CONV_UNIT_SLABS - 1) / TACO_CONV_UNIT_SLABS;
  if (units < 2 || base_tiles < 1) return 1;
  const size_t wgs = (size_t)base_tiles * units;
  // up to two workgroups per CU (what a CU's LDS holds at once): postnet at 400 / 500 frames 137 -> 114 / 116 us; beyond
  // that the split is slower than one workgroup per tile (600 frames: 155 against 143 us, 1000: 179 against 144)
  return (wgs <= (size_t)split_cap && wgs <= 2 * (size_t)n_cu) ? units : 1;
}

int taco_launch_conv(const float* in, float* out, const float* w, const float* bias, const float* alpha,
                     const float* beta, const float* residual, const int32_t* lens, int B, int c_in,
                     int c_out, int k, int T, int act, int layout, float* split_ws, int split_cap, int n_cu, hipStream_t s) {
  // the kernel addresses one utterance's input and the weights through 32-bit buffer offsets
  if ((uint64_t)c_in * (uint64_t)T * 4u >= 0xffffffffull || (uint64_t)c_out * (uint64_t)c_in * (uint64_t)k * 4u >= 0xffffffffull)
    TACO_FAIL(TACO_EINVAL, "convolution operand of %d x %d columns / %d x %d x %d weights exceeds 4 GiB", c_in, T, c_out, c_in, k);
  const int tiles_t = (T + CV_TT - 1) / CV_TT, tiles_co = (c_out + CV_CO - 1) / CV_CO;
  const int ci = taco_conv_ci(k);
  const int split = split_ws ? taco_conv_split_factor(tiles_t * tiles_co * B, (c_in + ci - 1) / ci, n_cu, split_cap) : 1;
  dim3 grid(tiles_t * split, tiles_co, B);
  if (split == 1 && tiles_t <= 65535) {          // (a grid's y extent ends at 65535)
    grid = dim3(tiles_co, tiles_t, B);
    layout |= 4;
  }
  switch (k) {
#define TACO_CONV_CASE(KS)                                                                          \
  case KS:                                                                                          \
    if (layout & 1)                                                                                 \
      hipLaunchKernelGGL((taco_conv_kernel<KS, true>), grid, dim3(256), taco_conv_lds_bytes(KS), s, in, out, w, bias, alpha, beta, \
                         residual, lens, c_in, c_out, T, act, layout, split, split_ws, split_cap); \
    else                                                                                            \
    hipLaunchKernelGGL((taco_conv_kernel<KS, false>), grid, dim3(256), taco_conv_lds_bytes(KS), s, in, out, w, bias, alpha, beta,  \
                       residual, lens, c_in, c_out, T, act, layout, split, split_ws, split_cap); \
    break;
    TACO_CONV_CASE(1)
    TACO_CONV_CASE(3)
    TACO_CONV_CASE(5)
    TACO_CONV_CASE(7)
    TACO_CONV_CASE(9)
#undef TACO_CONV_CASE
    default:
      TACO_FAIL(TACO_EINVAL, "convolution kernel size %d is not one of 1,3,5,7,9", k);
  }
  TACO_HIP(hipGetLastError());
  return TACO_OK;
}

int taco_launch_enc_lstm(taco_handle h, const float* gx, const int32_t* lens, int B, int T, float* memory,
                         hipStream_t s) {
  const taco_dims& d = h->d;
  const int n_wg_dir = (d.H_e + EL_UNITS - 1) / EL_UNITS;
  if (d.H_e > EL_HMAX || 2 * n_wg_dir > h->n_cu)
    TACO_FAIL(TACO_EINVAL, "encoder LSTM width %d exceeds the recurrence kernel's capacity %d", d.H_e, EL_HMAX);
  void (*kern)(EncLstmArgs) = d.H_e <= 128 ? taco_enc_lstm_kernel<4> : d.H_e <= 256 ? taco_enc_lstm_kernel<8>
                              : d.H_e <= 384 ? taco_enc_lstm_kernel<12> : taco_enc_lstm_kernel<16>;
  // Utterances per launch: all workgroups of a launch must be on the device together (the workgroups of an
  // utterance wait for each other), so no more than the device holds at once.
  if (h->enc_group < 1) {
    int per_cu = 0;
    TACO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 1024, 0));
    if (per_cu < 1) TACO_FAIL(TACO_EINVAL, "encoder recurrence kernel does not fit on a CU");
    h->enc_group = (h->n_cu * per_cu) / (2 * n_wg_dir);
  }
  const int cap = (int)(h->enc_gran_count / ((size_t)2 * EL_GRAN_DIR));
  const int group = h->enc_group < cap ? h->enc_group : cap;
  if (group < 1) TACO_FAIL(TACO_EINVAL, "encoder hand-off buffer too small");
  for (int b0 = 0; b0 < B; b0 += group) {
    const int n = B - b0 < group ? B - b0 : group;
    EncLstmArgs a;
    a.w_hh[0] = h->w.enc_lstm_w_hh[0];
    a.w_hh[1] = h->w.enc_lstm_w_hh[1];
    a.gx = gx + (size_t)b0 * 8 * d.H_e * T;
    a.memory = memory + (size_t)b0 * T * d.M;
    a.lens = lens; a.b = b0;
    a.H = d.H_e; a.T = T; a.M = d.M; a.n_wg_dir = n_wg_dir;
    a.gran = h->enc_gran; a.ctl = h->enc_ctl;
    TACO_HIP(hipMemsetAsync(h->enc_gran, 0, (size_t)n * 2 * EL_GRAN_DIR * sizeof(taco_granule), s));
    hipLaunchKernelGGL(kern, dim3(2 * n_wg_dir, n), dim3(1024), 0, s, a);
    TACO_HIP(hipGetLastError());
  }
  return TACO_OK;
}

// Hand-off words for as many utterances as a launch can hold: each needs at least two workgroups.
size_t taco_enc_lstm_granules(int n_cu) { return (size_t)(n_cu > 2 ? n_cu / 2 : 1) * 2 * EL_GRAN_DIR; }

int taco_launch_memory_tail(taco_handle h, const int64_t* speakers, const int32_t* lens, int B, int T,
                            float* memory, float* pm, hipStream_t s) {
  const taco_dims& d = h->d;
  if (d.M > TACO_NM * TACO_CHUNK) TACO_FAIL(TACO_EINVAL, "memory width %d too large", d.M);
  if (d.S > 0 && !speakers) TACO_FAIL(TACO_EINVAL, "speakers is NULL but the checkpoint models speakers");
  TailArgs a;
  a.speakers = speakers; a.spk_table = h->w.spk_table; a.lens = lens;
  a.memory = memory; a.T = T; a.M = d.M; a.S = d.S; a.H2 = 2 * d.H_e;
  a.spk_rows = d.spk_rows; a.bad_id = h->enc_ctl + 2;
  hipLaunchKernelGGL(taco_memory_tail_kernel, dim3(T, B), dim3(256), 0, s, a);
  TACO_HIP(hipGetLastError());
  // processed_memory = memory_layer(memory): a kernel-size-1 convolution, position-major in and out
  return taco_launch_conv(memory, pm, h->w.memory_w, nullptr, nullptr, nullptr, nullptr, lens, B, d.M, d.A, 1, T, 0,
                          /*position-major in and out*/ 3, h->conv_split, TACO_CONV_SPLIT_CAP, h->n_cu, s);
}
