// 3x3 stride-1 convolutions with a handful of OUTPUT channels (<= 4) on the vector ALUs: the data-gradient that reaches the image
// (`inc.0`: 16 -> 1 channels at C2, 64 -> 3 at C4; encoder_decoder.py:441-445 backward).  On the matrix cores such a layer costs as much as a 16-column one
// (the MFMA tile is 16 wide: 54.8 us at 16 -> 1 @16x256x256, 15/16 of the products wasted); as 9*Cin*Cout FMAs per pixel it is a pure streaming kernel:
// read the gradient (and, with the BatchNorm-backward prologue, the raw conv output), write Cout planes.
//
// Workgroup = 256 threads = a 16 x 64 pixel tile, 4 consecutive pixels per thread; input channels are staged 4 at a time through LDS (18 x 72 window per
// channel, prologue al*g + be*u + de applied while staging, zeros outside the image AFTER the prologue = the conv's zero padding), double buffered; a thread
// reads its 3 x 6 window per channel as one ds_read_b128 + two ds_read_b32 per row.  Weights: the data-gradient packed layout [tap][cin_pad][cout_pad], read
// through uniform (scalar) loads.
#include <algorithm>
#include "ms_conv_kernel.h"
#include "maxstyle_hip.h"

namespace ms {

constexpr int kSmTH = 16, kSmTW = 64, kSmCK = 2, kSmIH = kSmTH + 2, kSmRS = kSmTW + 8, kSmPS = kSmIH * kSmRS;       // window columns x0-4 .. x0+67
typedef unsigned sm_u32x4_t __attribute__((ext_vector_type(4)));
constexpr int kSmWS = 12;                                   // LDS floats per (channel, output channel) of a chunk's weights: 9 taps, padded to three 16-byte reads

// Second form (round 4): the loop body of the first one was ~560 instructions per 2-channel chunk for 72 multiply-adds per thread - 18 dependent scalar weight loads
// each with its own wait, a branch nest per staged quad, zero-filled staging registers, multiply and add as two instructions.  Now: the staging quads are buffer loads
// whose out-of-image offsets read as zero (no branches, no zero fill; fp32 storage), their addresses and LDS slots are computed once per thread, the chunk's weights
// travel with the data (one element per thread of the first waves, parked in LDS beside the window, read back as broadcast 16-byte reads), the prologue coefficients are
// uniform loads issued a chunk ahead, and the multiply-adds are FMAs (~190 instructions per chunk).  Same sums in the same order (channel by channel, tap by tap).
template <int COUT, bool PRO2, typename AT = float>
__global__ __launch_bounds__(256) void conv3x3_small_cout_kernel(const void* __restrict__ in, const void* __restrict__ in2, void* __restrict__ out,
                                                                 const float* __restrict__ w, const float* __restrict__ pro_a, const float* __restrict__ pro_b,
                                                                 const float* __restrict__ pro_c, int pro_cstride, int Cin, int H, int W, int cin_pad, int cout_pad,
                                                                 int tiles_x, int tiles_per_img) {
  __shared__ __attribute__((aligned(16))) float smem[2][kSmCK * kSmPS];
  __shared__ __attribute__((aligned(16))) float wsm[2][kSmCK * COUT * kSmWS];
  const int tid = threadIdx.x;
  // work item = (image, tile), numbered so that the workgroups of one XCD (every 8th workgroup id) take CONSECUTIVE tiles: neighbouring tiles share their halo rows and the
  // 128-byte lines their 288-byte row segments straddle, and the L2 is per XCD - with round-robin numbering those lines came over the fabric twice (counters: 257 MB
  // fetched for 138 MB at config 2, L2 hit rate 9 %)
  const int total = (int)gridDim.x;
  const int work = (total % 8 == 0) ? (int)(blockIdx.x % 8) * (total / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int n = work / tiles_per_img, tile = work % tiles_per_img;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int y0 = ty * kSmTH, x0 = tx * kSmTW;
  const size_t plane = (size_t)H * W;
  using IO = ActIO<AT>;                     // storage type of in / in2 / out: float | ms_bf16
  constexpr bool F32 = (IO::kBytes == 4);
  const size_t in_n = (size_t)n * Cin * plane;
  constexpr int ITEMS = kSmCK * kSmIH * (kSmRS / 4);        // float4 items per chunk: 2 x 18 x 18 = 648
  constexpr int NI = (ITEMS + 255) / 256;
  struct Stage { float4 rg[NI], ru[PRO2 ? NI : 1]; float rw, ca[kSmCK], cb[kSmCK], cd[kSmCK]; };      // one chunk on its way: data quads, its weight element, its coefficients
  Stage sa;                                                   // (two chunks in flight - a second Stage, 128 registers - measured no faster: 45.2 against 44.0 us at config 2, 250 against 246 at config 4)
  const int nchunks = (Cin + kSmCK - 1) / kSmCK;

  // per staging item of this thread (the same for every chunk): channel within the chunk, LDS slot (floats; -1: none), inside the image?, offset inside a channel plane
  int ic[NI], ilds[NI], ioff[NI]; bool iok[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int it = tid + j * 256;
    const int f = it % (kSmRS / 4), row = it / (kSmRS / 4);
    const int r = row % kSmIH, c = row / kSmIH;
    const int Y = y0 - 1 + r, X = x0 - 4 + 4 * f;
    ic[j] = c;
    ilds[j] = (it < ITEMS) ? c * (kSmPS / 4) + r * (kSmRS / 4) + f : -1;       // in 16-byte slots: the compiler then knows the alignment (ds_write_b128)
    iok[j] = (it < ITEMS) && Y >= 0 && Y < H && X >= 0 && X < W;          // W % 4 == 0: a quad is inside or outside as a whole
    ioff[j] = Y * W + X;
  }
  // the weight element this thread carries per chunk: (channel wc of the chunk, output channel wo, tap wt)
  const bool wthr = tid < kSmCK * COUT * 9;
  const int wc = tid / (COUT * 9), wo = (tid / 9) % COUT, wt = tid % 9;
  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(in) + in_n * IO::kBytes), 0, (int)((unsigned)Cin * (unsigned)plane * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(PRO2 ? in2 : in) + in_n * IO::kBytes), 0, (int)((unsigned)Cin * (unsigned)plane * 4u), 0x00020000);
  unsigned voff[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) voff[j] = iok[j] ? ((unsigned)ic[j] * (unsigned)plane + (unsigned)ioff[j]) * 4u : 0x80000000u;      // outside the image: an offset the resource refuses -> zeros

  auto load_chunk = [&](Stage& g, int c0) {
    if constexpr (F32) {
      const int soff = (int)((unsigned)c0 * (unsigned)plane * 4u);       // channels >= Cin lie behind the resource's last byte: zeros as well
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const sm_u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r1, (int)voff[j], soff, 0);
        g.rg[j] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        if (PRO2) {
          const sm_u32x4_t u = __builtin_amdgcn_raw_buffer_load_b128(r2, (int)voff[j], soff, 0);
          g.ru[j] = make_float4(__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3]));
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        g.rg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (PRO2) g.ru[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iok[j] && c0 + ic[j] < Cin) {
          const size_t off = in_n + (size_t)(c0 + ic[j]) * plane + (size_t)ioff[j];
          g.rg[j] = IO::ld4(in, off);
          if (PRO2) g.ru[j] = IO::ld4(in2, off);
        }
      }
    }
    g.rw = (wthr && c0 + wc < Cin) ? w[((size_t)wt * cin_pad + (c0 + wc)) * cout_pad + wo] : 0.f;
    if (PRO2) {
#pragma unroll
      for (int c = 0; c < kSmCK; ++c) {
        const bool in_c = (c0 + c < Cin);                                 // uniform
        g.ca[c] = in_c ? pro_a[(c0 + c) * pro_cstride] : 0.f; g.cb[c] = in_c ? pro_b[(c0 + c) * pro_cstride] : 0.f; g.cd[c] = in_c ? pro_c[(c0 + c) * pro_cstride] : 0.f;
      }
    }
  };
  auto store_chunk = [&](const Stage& g, int b) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      float4 v = g.rg[j];
      if (PRO2) {
        float al = g.ca[0], be = g.cb[0], de = g.cd[0];
#pragma unroll
        for (int c = 1; c < kSmCK; ++c) { al = (ic[j] == c) ? g.ca[c] : al; be = (ic[j] == c) ? g.cb[c] : be; de = (ic[j] == c) ? g.cd[c] : de; }
        de = iok[j] ? de : 0.f;                                           // zeros outside the image AFTER the prologue = the conv's zero padding (the loaded values are 0 there)
        v.x = al * v.x + (be * g.ru[j].x + de); v.y = al * v.y + (be * g.ru[j].y + de);
        v.z = al * v.z + (be * g.ru[j].z + de); v.w = al * v.w + (be * g.ru[j].w + de);
      }
      if (ilds[j] >= 0) reinterpret_cast<float4*>(smem[b])[ilds[j]] = v;
    }
    if (wthr) wsm[b][(wc * COUT + wo) * kSmWS + wt] = g.rw;
  };

  const int py = tid >> 4, px4 = (tid & 15) * 4;          // this thread's pixels: row y0+py, columns x0+px4 .. +3
  float acc[COUT][4];
#pragma unroll
  for (int o = 0; o < COUT; ++o)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[o][e] = 0.f;

  auto compute = [&](int b, int ch) {
#pragma unroll
    for (int c = 0; c < kSmCK; ++c) {
      const int ci = ch * kSmCK + c;
      if (ci < Cin) {
        float wv[COUT][kSmWS];
#pragma unroll
        for (int o = 0; o < COUT; ++o)
#pragma unroll
          for (int q = 0; q < kSmWS / 4; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(&wsm[b][(c * COUT + o) * kSmWS + 4 * q]);        // the same address in every lane: a broadcast read
            wv[o][4 * q] = t.x; wv[o][4 * q + 1] = t.y; wv[o][4 * q + 2] = t.z; wv[o][4 * q + 3] = t.w;
          }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          // window columns x-1 .. x+4 = LDS columns px4+3 .. px4+8, read as the three ALIGNED quads px4 .. px4+11: 16-byte reads at a 16-byte lane stride are
          // conflict-free, the dword pairs the compiler made of {q[3], quad, q[8]} were 8-way conflicts (counters: 70 % of the LDS cycles of the first form)
          const float4* q4 = reinterpret_cast<const float4*>(smem[b]) + (c * (kSmPS / 4) + (py + ky) * (kSmRS / 4) + (tid & 15));
          float4 lft = q4[0], mid = q4[1], rgt = q4[2];
          asm volatile("" : "+v"(lft.x), "+v"(lft.y), "+v"(lft.z), "+v"(lft.w));      // (every component "used": the compiler must not narrow the quads back to dword pairs)
          asm volatile("" : "+v"(rgt.x), "+v"(rgt.y), "+v"(rgt.z), "+v"(rgt.w));
          const float win[6] = {lft.w, mid.x, mid.y, mid.z, mid.w, rgt.x};
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int o = 0; o < COUT; ++o)
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[o][e] = __builtin_fmaf(wv[o][ky * 3 + kx], win[e + kx], acc[o][e]);
        }
      }
    }
  };
  load_chunk(sa, 0);
  for (int ch = 0; ch < nchunks; ++ch) {
    const int b = ch & 1;
    store_chunk(sa, b);
    if (ch + 1 < nchunks) load_chunk(sa, (ch + 1) * kSmCK);
    __syncthreads();
    compute(b, ch);
    // (the next iteration writes the OTHER buffers; the barrier of that iteration orders it against these reads)
  }
  const int y = y0 + py, x = x0 + px4;
  if (y < H && x < W) {
#pragma unroll
    for (int o = 0; o < COUT; ++o)
      IO::st4(out, ((size_t)n * COUT + o) * plane + (size_t)y * W + x, make_float4(acc[o][0], acc[o][1], acc[o][2], acc[o][3]));
  }
}


// ---------------------------------------------------------------------------------------------------------------------------------------------------------
// 3x3 stride-1 convolutions with a handful of INPUT channels (<= 4) and 16 output channels on the vector ALUs: the encoder's first conv on the image
// (`inc.0`: 1 -> 16 channels at C2; encoder_decoder.py:441-445 forward).  On the matrix cores the layer pads its single input channel to an 8-channel chunk
// (18 MFMAs per 16 pixels of which 3 carry data) and goes through the whole staging pipeline for a 4 MB input: 34 us for what is a 67 MB write.  Here a
// thread computes 4 pixels x 16 channels from a 3 x 6 window per input channel read straight from global memory (the input is L2-resident; neighbouring
// threads share their windows in the vector L1), weights through uniform (scalar) loads, 16 coalesced 16-byte stores - and keeps the per-channel running
// (count, mean, M2) of its outputs in registers (Chan), merged once per workgroup into the conv kernels' statistics table (one slot per workgroup, header
// {slots, launch epoch}: what ms_bn_finalize and the `_xfin` consumers read).  Persistent grid (<= 2 workgroups per CU), tiles of 16 x 64 pixels.
constexpr int kSciCout = 16;

__device__ __forceinline__ void sci_chan_merge(float& n, float& a, float& b, float nb, float ab, float bb) {      // (n, mean a, M2 b) += (nb, ab, bb)
  const float nn = n + nb;
  const float w = (nn > 0.f) ? nb / nn : 0.f;
  const float d = ab - a;
  a += d * w; b += bb + d * d * n * w; n = nn;
}

template <int CIN, typename AT = float>
__global__ __launch_bounds__(256) void conv3x3_small_cin_kernel(const void* __restrict__ in, void* __restrict__ out, const float* __restrict__ w,
                                                                const float* __restrict__ bias, float4* __restrict__ stats, int N, int H, int W, int cin_pad,
                                                                int cout_pad, int tiles_x, int tiles_per_img) {
  using IO = ActIO<AT>;
  __shared__ float red[4][kSciCout][3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int py = tid >> 4, px4 = (tid & 15) * 4;
  const size_t plane = (size_t)H * W;
  const int ntiles = N * tiles_per_img;
  float st_n = 0.f, st_mean[kSciCout], st_m2[kSciCout];
#pragma unroll
  for (int o = 0; o < kSciCout; ++o) { st_mean[o] = 0.f; st_m2[o] = 0.f; }
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int n = t / tiles_per_img, tile = t - n * tiles_per_img;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int y = ty * kSmTH + py, x = tx * kSmTW + px4;
    const bool live = (y < H) && (x < W);                   // W % 4 == 0: the quad is inside or outside as a whole
    float acc[kSciCout][4];
#pragma unroll
    for (int o = 0; o < kSciCout; ++o) {
      const float bv = bias != nullptr ? bias[o] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[o][e] = bv;
    }
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
      const size_t pb = ((size_t)n * CIN + ci) * plane;
      float win[3][6];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int Y = y - 1 + ky;
        const bool rok = live && Y >= 0 && Y < H;
        const size_t ro = pb + (size_t)(rok ? Y : 0) * W;
        const float4 mid = rok ? IO::ld4(in, ro + x) : make_float4(0.f, 0.f, 0.f, 0.f);
        win[ky][0] = (rok && x > 0) ? IO::ld1(in, ro + x - 1) : 0.f;
        win[ky][1] = mid.x; win[ky][2] = mid.y; win[ky][3] = mid.z; win[ky][4] = mid.w;
        win[ky][5] = (rok && x + 4 < W) ? IO::ld1(in, ro + x + 4) : 0.f;
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          // uniform address: 16 consecutive weights, scalar loads, hoisted out of the tile loop by the compiler (144 SGPRs, part of them parked in vector-register lanes).
          // Measured alternatives (profiles/r03_experiments.txt 18): weights in LDS 118 us (hoisted into 144 VECTOR registers: spills), re-read per tile through the
          // scalar cache 38 us, packed v_pk_fma_f32 31.5 us - this form 30.4 us.
          const float* wr = w + ((size_t)(ky * 3 + kx) * cin_pad + ci) * cout_pad;
#pragma unroll
          for (int o = 0; o < kSciCout; ++o) {
            const float wv = wr[o];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[o][e] = __builtin_fmaf(wv, win[ky][e + kx], acc[o][e]);      // (explicit: the library is built with -ffp-contract=off)
          }
        }
    }
    if (live) {
#pragma unroll
      for (int o = 0; o < kSciCout; ++o)
        IO::st4(out, ((size_t)n * kSciCout + o) * plane + (size_t)y * W + x, make_float4(acc[o][0], acc[o][1], acc[o][2], acc[o][3]));
      if (stats != nullptr) {
        // the quad's (4, mean, M2) per channel, Chan-merged into the thread's running statistics (of the values as STORED: bf16 storage rounds first)
        const float nt_ = st_n + 4.f;
        const float wgt = 4.f / nt_;
#pragma unroll
        for (int o = 0; o < kSciCout; ++o) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) { if constexpr (IO::kBytes == 2) v[e] = IO::up(ms_to_bf16(acc[o][e])); else v[e] = acc[o][e]; }
          const float gm = ((v[0] + v[1]) + (v[2] + v[3])) * 0.25f;
          const float d0 = v[0] - gm, d1 = v[1] - gm, d2 = v[2] - gm, d3 = v[3] - gm;
          const float q = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
          const float dd = gm - st_mean[o];
          st_mean[o] += dd * wgt;
          st_m2[o] += q + dd * dd * st_n * wgt;
        }
        st_n = nt_;
      }
    }
  }
  if (stats == nullptr) return;
  // one slot per workgroup and channel: lanes of a wave (xor shuffles), then the four waves through LDS, in a fixed order
#pragma unroll
  for (int o = 0; o < kSciCout; ++o) {
    float n_ = st_n, a_ = st_mean[o], b_ = st_m2[o];
#pragma unroll
    for (int off = 1; off <= 32; off <<= 1) {
      const float nb = __shfl_xor(n_, off, 64), ab = __shfl_xor(a_, off, 64), bb = __shfl_xor(b_, off, 64);
      sci_chan_merge(n_, a_, b_, nb, ab, bb);
    }
    if (lane == 0) { red[wave][o][0] = n_; red[wave][o][1] = a_; red[wave][o][2] = b_; }
  }
  __syncthreads();
  if (tid < kSciCout) {
    float n_ = red[0][tid][0], a_ = red[0][tid][1], b_ = red[0][tid][2];
#pragma unroll
    for (int w_ = 1; w_ < 4; ++w_) sci_chan_merge(n_, a_, b_, red[w_][tid][0], red[w_][tid][1], red[w_][tid][2]);
    stats[1 + (size_t)tid * kStatSlots + blockIdx.x] = make_float4(n_, a_, b_, 0.f);
  }
  if (blockIdx.x == 0 && tid == 0) {      // header: {slots in use, launch epoch of this table}: the `_xfin` consumers' tag (conv_table_tail's convention)
    const unsigned ep = __float_as_uint(stats[0].y) + 1u;
    stats[0] = make_float4((float)gridDim.x, __uint_as_float(ep == 0u ? 1u : ep), 0.f, 0.f);
  }
}

}  // namespace ms

using namespace ms;

extern "C" int ms_conv3x3_small_cout_ok(int Cout, int W) { return (Cout >= 1 && Cout <= 4 && W % 4 == 0) ? 1 : 0; }

// out [N,Cout,H,W] = conv3x3(P(in), w) with Cout <= 4, stride 1, padding 1; w = packed weights [9][cin_pad][cout_pad] (for a data-gradient: the
// data-gradient layout).  pro_mode 0: P = identity; 2: P = pro_a[c]*in + pro_b[c]*in2 + pro_c[c] (BatchNorm backward, coefficient records of stride pro_cstride).
template <typename AT>
static int small_cout_impl(const void* in, const void* in2, void* out, const float* w_packed, int N, int Cin, int H, int W, int Cout,
                           int pro_mode, const float* pro_a, const float* pro_b, const float* pro_c, int pro_cstride, void* stream) {
  if (N < 1 || Cin < 1 || H < 1 || W < 1 || !ms_conv3x3_small_cout_ok(Cout, W)) { set_error("ms_conv3x3_small_cout: Cout <= 4, W %% 4 == 0"); return MS_ERR_INVALID; }
  if (pro_mode != 0 && pro_mode != 2) { set_error("ms_conv3x3_small_cout: pro_mode 0 or 2"); return MS_ERR_INVALID; }
  if (pro_mode == 2 && (in2 == nullptr || pro_a == nullptr || pro_b == nullptr || pro_c == nullptr)) { set_error("ms_conv3x3_small_cout: prologue operands missing"); return MS_ERR_INVALID; }
  if (!aligned16(in) || !aligned16(out) || (in2 != nullptr && !aligned16(in2))) { set_error("ms_conv3x3_small_cout: tensors must be 16-byte aligned"); return MS_ERR_ALIGN; }
  if ((size_t)Cin * H * W * 4u >= ((size_t)1 << 31)) { set_error("ms_conv3x3_small_cout: one image of the input must stay below 2 GiB (buffer-resource addressing)"); return MS_ERR_INVALID; }
  const int tiles_x = cdiv(W, kSmTW), tiles_y = cdiv(H, kSmTH);
  const int cin_pad = (Cin + 3) / 4 * 4, cout_pad = (Cout + 63) / 64 * 64;
  if ((long)tiles_x * tiles_y * N > 0x7fffffffL) { set_error("ms_conv3x3_small_cout: too many tiles"); return MS_ERR_INVALID; }
  dim3 grid(tiles_x * tiles_y * N), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int cs = pro_cstride < 1 ? 1 : pro_cstride;
#define MS_SM(CO, P2) MS_LAUNCH((conv3x3_small_cout_kernel<CO, P2, AT>), grid, block, 0, st, in, in2, out, w_packed, pro_a, pro_b, pro_c, cs, Cin, H, W, cin_pad, cout_pad, tiles_x, tiles_x * tiles_y)
  if (pro_mode == 2) { switch (Cout) { case 1: MS_SM(1, true); break; case 2: MS_SM(2, true); break; case 3: MS_SM(3, true); break; default: MS_SM(4, true); } }
  else { switch (Cout) { case 1: MS_SM(1, false); break; case 2: MS_SM(2, false); break; case 3: MS_SM(3, false); break; default: MS_SM(4, false); } }
#undef MS_SM
  return check_launch("conv3x3_small_cout");
}
extern "C" int ms_conv3x3_small_cout(const float* in, const float* in2, float* out, const float* w_packed, int N, int Cin, int H, int W, int Cout,
                                     int pro_mode, const float* pro_a, const float* pro_b, const float* pro_c, int pro_cstride, void* stream) {
  return small_cout_impl<float>(in, in2, out, w_packed, N, Cin, H, W, Cout, pro_mode, pro_a, pro_b, pro_c, pro_cstride, stream);
}
// bf16 activation storage (in, in2, out are bf16 bit patterns): see the bf16 section of include/maxstyle_hip.h
extern "C" int ms_conv3x3_small_cout_bf16(const uint16_t* in, const uint16_t* in2, uint16_t* out, const float* w_packed, int N, int Cin, int H, int W, int Cout,
                                          int pro_mode, const float* pro_a, const float* pro_b, const float* pro_c, int pro_cstride, void* stream) {
  return small_cout_impl<ms_bf16>(in, in2, out, w_packed, N, Cin, H, W, Cout, pro_mode, pro_a, pro_b, pro_c, pro_cstride, stream);
}

// 1: the entry point takes the shape AND is the faster choice (Cin = 1: rows of whole 16-pixel M-tiles take the taps-as-K matrix form - ms_conv2d's bits; other
// rows the vector form, 30 vs 34 us at 16x1x256x256, ANOTHER rounding; with 2..4 input channels the vector kernel is correct but its 300..600 weights no longer fit the
// scalar registers); 0: use ms_conv2d
extern "C" int ms_conv3x3_small_cin_ok(int Cin, int Cout, int W) { return (Cin == 1 && Cout == kSciCout && W % 4 == 0) ? 1 : 0; }

// out [N,16,H,W] = conv3x3(in [N,Cin,H,W], w) + bias, Cin <= 4, stride 1, padding 1; w = packed forward weights [9][cin_pad][cout_pad] (ms_conv2d's layout).
// stats (may be NULL): the statistics table of ms_conv2d ([1 + 16 * ms_conv_stats_parts()] float4, header {slots, epoch}) for ms_bn_finalize / the `_xfin` consumers.
// Third form (round 5), Cin = 1: the nine TAPS are the K dimension of the matrix instruction.  A lane (m, k) of MFMA j holds tap 4 j + k of pixel m of a 16-pixel
// M-tile - read straight from the (L2-resident, 4 MB at config 2) image with buffer loads whose out-of-image offsets return the conv's zero padding - against the
// weight of that tap for output channel m: 3 MFMAs per 16 pixels x 16 channels (the padded 8-channel chunk of the general kernels issues 18, the vector form above
// ~800 FMAs per thread).  No LDS, no staging waves: a wave walks strips of 64 pixels of one image row (4 M-tiles: 12 loads, 12 MFMAs, 4 x 16-byte stores per lane),
// keeps the per-lane running (count, mean, M2) of its output channel and ends in the conv kernels' table tail (one slot per workgroup).  What remains is the 67 MB
// write.  The matrix instruction adds its four K products to the accumulator one after the other, in K order (measured: the output has the BITS of the general kernels,
// which feed the same taps one per instruction - tests/test_k9_gpu.py), so the layer's rounding does not depend on which kernel ran it.
__global__ __launch_bounds__(256) void conv3x3_k9_kernel(const ConvArgs a, int nsx) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 3];
  constexpr int NJ = 3;
  constexpr int OOB = (int)0x80000000;
  const int wave = MS_TID >> 6, lane = MS_TID & 63, m = lane & 15, k = lane >> 4;
  const int H = a.Hs, W = a.Ws;
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)((unsigned)a.N * H * W * 4u), 0x00020000);
  // lane constants: tap of MFMA j, its weight, its (dy, dx)
  float bw[NJ];
  int dy[NJ], dx[NJ], loff[NJ];
  bool tv[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int tap = 4 * j + k;
    tv[j] = tap < 9;
    const int tp = min(tap, 8);
    dy[j] = tp / 3 - 1; dx[j] = tp % 3 - 1;
    loff[j] = (dy[j] * W + dx[j] + m) * 4;
    bw[j] = tv[j] ? a.w[(size_t)tp * a.cin_pad * a.cout_pad + m] : 0.f;
  }
  const float bias_v = (a.bias != nullptr) ? a.bias[m] : 0.f;
  const int nstrips = a.N * H * nsx;
  float st_n = 0.f, st_mean[1] = {0.f}, st_m2[1] = {0.f};
  for (int s = (int)blockIdx.x * 4 + wave; s < nstrips; s += (int)gridDim.x * 4) {
    const int sx = s % nsx, row = s / nsx;              // row = n * H + y
    const int y = row % H;
    const int x0 = sx * 64;
    const int base = (row * W + x0) * 4;
    float av[4][NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const bool rok = tv[j] && ((unsigned)(y + dy[j]) < (unsigned)H);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bool ok = rok && ((unsigned)(x0 + 16 * t + m + dx[j]) < (unsigned)W);
        av[t][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_in, ok ? (base + 64 * t + loff[j]) : OOB, 0, 0));
      }
    }
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][j], bw[j], acc[t], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] += bias_v;
    }
    // D layout: this lane holds output channel m of pixels x0 + 16 t + 4 k .. + 3 (W % 16 == 0: an M-tile is inside the row or outside it)
    const int nt_ok = min(4, (W - x0) >> 4);
    if (a.stats != nullptr) {
      const float cnt = 4.f * (float)nt_ok;
      const float rc = __builtin_amdgcn_rcpf(cnt);
      const float ntot = st_n + cnt;
      const float wgt = cnt * __builtin_amdgcn_rcpf(ntot);
      float sum = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) sum += (t < nt_ok) ? acc[t][r] : 0.f;
      const float mean = sum * rc;
      float q = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = acc[t][r] - mean; q += (t < nt_ok) ? d * d : 0.f; }
      const float d = mean - st_mean[0];
      st_mean[0] += d * wgt;
      st_m2[0] += q + d * d * st_n * wgt;
      st_n = ntot;
    }
    const int n = row / H;
    float* op = a.out + ((size_t)(n * 16 + m) * H + y) * W + x0 + 4 * k;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (t < nt_ok) *reinterpret_cast<float4*>(op + 16 * t) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
  }
  if (a.stats != nullptr) conv_table_tail<1, true>(a, red, (int)blockIdx.x, 1, st_n, st_mean, st_m2);
}

static int conv_k9_launch(const float* in, float* out, const float* w_packed, const float* bias, int N, int H, int W, float* stats, hipStream_t st) {
  ConvArgs a{};
  a.in = in; a.out = out; a.w = w_packed; a.bias = bias; a.stats = stats;
  a.N = N; a.Cin = 1; a.Hs = H; a.Ws = W; a.Hin = H; a.Win = W; a.Cout = kSciCout; a.Hout = H; a.Wout = W;
  a.cin_pad = 4; a.cout_pad = 64; a.cout_real = kSciCout; a.ncb = 1;
  const int nsx = cdiv(W, 64);
  const long nstrips = (long)N * H * nsx;
  const int grid = (int)std::min<long>(cdiv(nstrips, 4L), std::min<long>(8L * num_cus(), kStatSlots));
  MS_LAUNCH(conv3x3_k9_kernel, dim3(grid), dim3(256), 0, st, a, nsx);
  return check_launch("conv3x3_k9");
}

template <typename AT>
static int small_cin_impl(const void* in, void* out, const float* w_packed, const float* bias, int N, int Cin, int H, int W, int Cout, float* stats, void* stream) {
  if (N < 1 || H < 1 || W < 1 || Cin < 1 || Cin > 4 || Cout != kSciCout || W % 4 != 0) { set_error("ms_conv3x3_small_cin: Cin <= 4, Cout == 16, W %% 4 == 0"); return MS_ERR_INVALID; }
  if (!aligned16(in) || !aligned16(out) || (stats != nullptr && !aligned16(stats))) { set_error("ms_conv3x3_small_cin: tensors must be 16-byte aligned"); return MS_ERR_ALIGN; }
  hipStream_t st = (hipStream_t)stream;
  if constexpr (sizeof(AT) == 4) {           // Cin = 1, rows of whole M-tiles, fp32 storage: the taps-as-K matrix form (option "conv.k9"; 0: the vector form)
    if (Cin == 1 && W % 16 == 0 && opt(OPT_CONV_K9) != 0 && (long long)N * H * W * 4 < (1LL << 31) && aligned16(w_packed))
      return conv_k9_launch((const float*)in, (float*)out, w_packed, bias, N, H, W, stats, st);
  }
  const int tiles_x = cdiv(W, kSmTW), tiles_y = cdiv(H, kSmTH);
  const long ntiles = (long)N * tiles_x * tiles_y;
  const int cin_pad = (Cin + 3) / 4 * 4, cout_pad = (Cout + 63) / 64 * 64;
  const int grid = (int)std::min<long>(ntiles, std::min<long>((Cin == 1 ? 3L : 2L) * num_cus(), kStatSlots));      // resident workgroups per CU: 143 VGPRs at Cin = 1
#define MS_SCI(CI) MS_LAUNCH((conv3x3_small_cin_kernel<CI, AT>), dim3(grid), dim3(256), 0, st, in, out, w_packed, bias, (float4*)stats, N, H, W, cin_pad, cout_pad, tiles_x, tiles_x * tiles_y)
  switch (Cin) { case 1: MS_SCI(1); break; case 2: MS_SCI(2); break; case 3: MS_SCI(3); break; default: MS_SCI(4); }
#undef MS_SCI
  return check_launch("conv3x3_small_cin");
}
extern "C" int ms_conv3x3_small_cin(const float* in, float* out, const float* w_packed, const float* bias, int N, int Cin, int H, int W, int Cout, float* stats, void* stream) {
  return small_cin_impl<float>(in, out, w_packed, bias, N, Cin, H, W, Cout, stats, stream);
}
extern "C" int ms_conv3x3_small_cin_bf16(const uint16_t* in, uint16_t* out, const float* w_packed, const float* bias, int N, int Cin, int H, int W, int Cout, float* stats, void* stream) {
  return small_cin_impl<ms_bf16>(in, out, w_packed, bias, N, Cin, H, W, Cout, stats, stream);
}
