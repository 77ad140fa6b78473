// "Wide-read" 3x3 stride-1 convolution kernel for gfx950: the second generation of conv_mfma_kernel (ms_conv_kernel.h) for the layers
// whose output rows are >= 64 pixels wide (the 256^2, 128^2 and 64^2 levels of the networks = most of the FLOPs of a step).
//
// What changed, and why (all measured on MI355X, see profiles/ and DESIGN.md):
//   * The first kernel reads every MFMA operand with its own ds_read_b32: 5 LDS reads per 4 MFMAs.  With eight MFMA waves per CU that is
//     ~62 % of the LDS pipe before the staging waves write anything - the consumers were LDS-bound, not MFMA-bound.  Here the MFMA M index
//     is BLOCKED over pixels: lane m of a 16-lane group owns pixels 4m..4m+3 of a 64-pixel row segment, one M-tile per pixel-in-quad.
//     The three kernel columns of four M-tiles then need SIX consecutive input floats per lane, channel and kernel row: one ds_read_b128 +
//     one ds_read_b64 feed 12*NT MFMAs (was: 12 + 3*NT ds_read_b32).  Plane stride == 0 (mod 64 dwords) keeps the b128 reads conflict-free.
//   * The accumulator fragment of a lane is then 16 CONSECUTIVE pixels of one output channel: the epilogue stores four 16-byte vectors per
//     channel block straight from registers and the BatchNorm statistics need no cross-lane shuffles beyond the 4 lane groups.
//   * Staging waves: everything tile-independent (LDS offsets, image-relative element offsets, per-channel prologue coefficients) is hoisted;
//     per tile an item costs one add, and masks / zero-fill selects only run for tiles on the image border (wave-uniform branch).  VALU
//     work on the staging waves competes with the MFMA issue of the wave sharing their SIMD, so it is kept minimal.
// Same contract as conv_mfma_kernel<3,1,FETCH_NORMAL,NT,true,false,IN2>: ConvArgs, packed weights, prologues, epilogues, statistics table.
#pragma once
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include "ms_conv_kernel.h"

namespace ms {

// R = output rows per MFMA wave: tile height TH = 4*R.  R = 2 (8-row tiles) stages 10 input rows for 8 output rows instead of 6 for 4 (halo overhead
// 1.29 instead of 1.55: 17 % less work for the staging waves, which bound the two-tensor variants), re-uses each A window for two output rows
// (4 window reads per channel group instead of 6) and each B fragment twice, and halves the barriers per FLOP; it needs enough tiles to fill the chip.
template <int NT, int PRO, int R = 1>
struct WideGeo {
  static constexpr int TH = 4 * R, TW = 64;
  // input channels per K-chunk: two stage buffers, two workgroups per CU; the two-tensor prologue stages twice the registers per channel
  static constexpr int CK = 8;   // measured: 8-channel chunks beat 16 for every variant (60.8 vs 62.1 us at 16->16 @256^2; the two-tensor prologue spills with 16)
  static constexpr int IH = TH + 2;
  static constexpr int RS = TW + 4;                           // LDS row: column 0 = left halo (x0-1), 1..64 interior, 65 = right halo
  static constexpr int PS = (IH * RS + 63) / 64 * 64;         // >= IH*RS (408 -> 448, 680 -> 704), == 0 (mod 64): conflict-free ds_read_b128 of the A windows
  static constexpr int WS = (NT == 1) ? 16 : NT * 16 + 16;    // weight row stride (bank-conflict-free B fragments)
  static constexpr int BUF = CK * PS + 9 * CK * WS;           // floats per stage buffer
  static constexpr int Q_ITEMS = CK * IH * (TW / 4);          // interior 16-byte items
  static constexpr int H_ITEMS = CK * IH * 2;                 // halo scalars
  static constexpr int NQI = (Q_ITEMS + 255) / 256, NHI = (H_ITEMS + 255) / 256;
  static constexpr int W_ITEMS = 9 * CK * (16 * NT / 4);
  static constexpr int NWI = (W_ITEMS + 255) / 256;
};

// Geometry of the bf16-MFMA mode (AT = ms_bf16m): v_mfma_f32_16x16x16_bf16 per (tap, pixel-in-quad, channel block) with 8-channel K-chunks - K groups 0 and 1 of
// the instruction carry the chunk's two channel quads, groups 2 and 3 read LDS planes that are zeroed once and never written (16-channel chunks need twice the
// staging registers: 292 bytes of scratch per lane with the two-tensor prologue, 218 registers with two channel blocks per lane; half-empty bf16 MFMAs still
// cost a quarter of the fp32 ones).  The LDS tile is
// CHANNEL-QUAD interleaved: entry (g, row, col) = the 4 bf16 values of channels 4g..4g+3 at that pixel (8 bytes), so that the A operand of a lane
// (pixel row m, K group g = lane >> 4: K = 4g..4g+3) for its 6-pixel window is 48 CONTIGUOUS bytes (three ds_read_b128 feed 12*NT MFMAs), and the B operand
// (weights, entry (tap, g, cout) = 4 channels) one ds_read_b64.  Offsets below are in BYTES.
template <int NT, int PRO>
struct WideGeoBF {
  static constexpr int TH = 4, TW = 64, CK = 8, IH = TH + 2;
  static constexpr int RSB = TW + 4;                              // entries per staged row: 0 = left halo, 1..64 interior, 65 = right halo
  static constexpr int GP = IH * RSB * 8;                          // bytes of one channel-quad plane (3264)
  static constexpr int A_BYTES = 4 * GP;
  static constexpr int B_BYTES = 9 * 4 * 16 * NT * 8;
  static constexpr int BUF = (A_BYTES + B_BYTES) / 4;              // floats per stage buffer
  static constexpr int RS = RSB, PS = GP / 4, WS = 16;             // (unused by the bf16-MFMA code paths; keep the shared constants defined)
  static constexpr int Q_ITEMS = CK * IH * (TW / 4), H_ITEMS = CK * IH * 2;
  static constexpr int NQI = (Q_ITEMS + 255) / 256, NHI = (H_ITEMS + 255) / 256;
  static constexpr int W_ITEMS = 9 * CK * (16 * NT / 4), NWI = (W_ITEMS + 255) / 256;
};

// Geometry of the Winograd mode (AT = ms_f32w): F(2x2, 3x3), one channel block per lane.  The staged input tile is the fp32 one (same rows / columns / halo);
// its plane stride is == 32 (mod 64 dwords) so that the 8-byte patch reads of the four K lanes groups of a wave fall on disjoint banks.  The weight region of a
// stage holds the chunk's TRANSFORMED weights U = G g G^T as [16 positions][CK channels][16 output channels]: a lane's B fragment of position p and channel
// group cg is one ds_read_b32 at p*128 + cg*64 + lane (64 consecutive dwords per wave: conflict-free).
// TW_ = 64: 4-row tiles (waves = 2 tile rows x 2 halves of 32 pixels); TW_ = 32: 8-row tiles (waves = 4 tile rows) for rows of 20..63 pixels.
// NT = 16-channel output blocks per staged input tile (round 4): the tile is staged, prologue'd and B^T d B-transformed ONCE for 16*NT output channels; the weight
// region holds [16 positions][NT blocks][CK channels][16 output channels] (a B fragment of block j = one ds_read_b32 at p*128*NT + j*128 + cg*64 + lane).
template <int NT_, int PRO, int TW_ = 64>
struct WideGeoW {
  static constexpr int TW = TW_, TH = 256 / TW_, CK = 8, IH = TH + 2;
  static constexpr int RS = TW + 4;
  static constexpr int PS = 416;                                  // >= IH*RS (6*68 = 408, 10*36 = 360), == 32 (mod 64)
  static_assert(IH * RS <= PS, "plane stride");
  static constexpr int WS = 16;
  static constexpr int UP = CK * 16 * NT_;                        // floats of one transform position in the weight region
  static constexpr int BUF = CK * PS + 16 * UP;                   // floats per stage buffer
  static constexpr int Q_ITEMS = CK * IH * (TW / 4), H_ITEMS = CK * IH * 2;
  static constexpr int NQI = (Q_ITEMS + 255) / 256, NHI = (H_ITEMS + 255) / 256;
  static constexpr int W_ITEMS = CK * 16 * NT_, NWI = 1;
  static_assert(W_ITEMS <= 256, "one (input channel, output channel) pair per staging thread");
};

// Block form of the Winograd mode (AT = ms_f32wb / ms_bf16wb): staging wave s stages patch s = block s of the work item (10 rows x 10 columns per channel, own halo),
// MFMA wave w multiplies block w: lane m = tile (m >> 2, m & 3) of the block's 4 x 4 tiles, so after the MFMAs a lane holds tile ROW k (2 output rows x 8 pixels) of
// channel m - the epilogue's 2 x 8 pixels per lane, unchanged.  Patch rows are 12 dwords, patches 120, the channel plane 480 == 32 (mod 64): the 8-byte patch reads of a
// 32-lane group (two K lane groups x 16 tiles) fall on 64 distinct banks.
template <int NT_, int PRO>
struct WideGeoWB {
  static constexpr int TW = 8, TH = 8, CK = 8, IH = TH + 2;
  static constexpr int RS = 12, PP = IH * RS, PS = 480;
  static_assert(4 * PP <= PS, "plane stride");
  static constexpr int WS = 16;
  static constexpr int UP = CK * 16 * NT_;
  static constexpr int BUF = CK * PS + 16 * UP;
  static constexpr int Q_ITEMS = CK * IH * (TW / 4), H_ITEMS = CK * IH * 2;      // per PATCH: one staging wave (64 lanes) stages one patch
  static constexpr int NQI = (Q_ITEMS + 63) / 64, NHI = (H_ITEMS + 63) / 64;
  static constexpr int W_ITEMS = CK * 16 * NT_, NWI = 1;
};

// Flat form of the Winograd mode (AT = ms_f32wf, round 6): images of W = 20 pixels per row (config 4's deepest levels).  The 2x2-output tiles of the whole
// batch form ONE list (image, tile row, tile column: 10 tiles per row); a work item = 64 consecutive tiles of it x 32 output channels, MFMA wave w multiplies tiles
// 16 w .. 16 w + 15 - every MFMA row is a real tile (the 8-row x 32-pixel tile fills 62 % x 83 % of its rows here).  Staged per chunk: the BAND of image rows the item's tiles
// touch - of one image, or (the list runs across images) the last rows of one and the first rows of the next: band row r = image row 2 ty0 - 1 + r of the first image up to
// and including the row BELOW it (zero), then rows -1 (zero), 0, 1 .. of the next.  64 tiles touch at most 8 tile rows in all: 16 pixel rows + 2 halo rows + 2 at an image
// boundary = 20 band rows of 22 floats (column 0 / 21 = the left / right halo: always outside the image, always zero); plane stride 480 == 32 (mod 64).
// A lane's 4 x 4 patch starts at an even column (8-byte reads), its base offset is per lane and per item (tile -> band row, column); after the MFMAs a lane holds four
// consecutive tiles of the list, stored tile by tile (8-byte stores).  Per output element the K loop is the tiled form's: the same bits in `out`.
template <int NT_, int PRO, int W_ = 20>
struct WideGeoWF {
  static constexpr int W = W_, TXR = W / 2;      // (W = 24 / 28: 64 tiles touch 7 / 6 tile rows at most - 18 / 16 band rows of 26 / 30 floats: the same 20-row, 480-float plane)
  static_assert(W == 20 || W == 24 || W == 28, "flat form: rows of 20 / 24 / 28 pixels");
  static constexpr int TW = W, TH = 4, CK = 8;
  static constexpr int IH = 2 * ((64 + TXR - 1 + TXR - 1) / TXR) + 4;      // band rows: 64 consecutive tiles starting anywhere in a tile row touch ceil((64 + TXR - 1) / TXR) tile rows, + 2 halo rows, + 2 at an image boundary: 20 / 18 / 16
  static constexpr int RS = W + 2, PS = 480;
  static_assert(IH * RS <= PS && PS % 64 == 32, "plane stride");
  static constexpr int WS = 16;
  static constexpr int UP = CK * 16 * NT_;
  static constexpr int BUF = CK * PS + 16 * UP;
  static constexpr int Q_ITEMS = CK * IH * (TW / 4), H_ITEMS = CK * IH * 2;      // (the halo items are never inside the image: they store the zeros of columns 0 / 21)
  static constexpr int NQI = (Q_ITEMS + 255) / 256, NHI = (H_ITEMS + 255) / 256;
  static constexpr int W_ITEMS = CK * 16 * NT_, NWI = 1;
};


// PRO: 0 none, 1 BatchNorm apply + LeakyReLU, 2 BatchNorm backward (two tensors)
// AF ("all full"): cin_pad is a multiple of the K-chunk - no ragged channel group anywhere in the layer, the guarded MFMA loop is not instantiated
// AT = storage type of the activation tensors (float | ms_bf16, ms_common.h ActIO)
// FX ("fixed launch facts", round 6): -1 = everything below is read from ConvArgs at run time (the generic instantiation: also the only one that honours the timing-only
// ablation bits a_dbg).  FX >= 0 = the epilogue kind (bits 0-2 = epi_mode), statistics (bit 3), bias (bit 4) and the Winograd appendix (bit 5) are COMPILE-TIME facts: the
// hot instantiations carry no branch on them.  Found with tools/isa_census.py on the dominant launch of the C2 step (two-tensor prologue + activation backward): with
// run-time flags the MFMA waves executed, per work item, 36 v_mov that zero the accumulators for the ablation path (hoisted in front of its branch), 16 bias additions +
// 16 v_mov selecting between the biased / unbiased transform (no bias in a data-gradient), the 64-bit address arithmetic of the border epilogue in front of the branch that
// skips it, and ~200 scalar instructions of mode dispatch.  Same arithmetic in the same order: bit-identical results (tests/test_wino_gpu.py).
constexpr int kFxStats = 8, kFxBias = 16, kFxWu = 32;
template <int NT, int PRO, int R, bool AF, typename AT = float, int FX = -1>
__global__ __launch_bounds__(512, (NT == 1 ? 4 : 2)) void conv_wide_kernel(const ConvArgs a) {
  constexpr bool FIXED = FX >= 0;
  const int a_dbg = FIXED ? 0 : a.dbg;
  const int a_epi = FIXED ? (FX & 7) : a.epi_mode;
  const bool a_has_stats = FIXED ? ((FX & kFxStats) != 0) : (a.stats != nullptr);
  const bool a_has_bias = FIXED ? ((FX & kFxBias) != 0) : (a.bias != nullptr);
  const bool a_has_wu = FIXED ? ((FX & kFxWu) != 0) : (a.wu != nullptr);
#ifdef MS_CONV_TRACE_BUILD
  // per-workgroup wall-clock stamps (s_memrealtime, 100 MHz): entry, first MFMA chunk, exit - where a launch spends the time outside its steady state (tools/trace_conv.py)
  if (a.trace != nullptr && threadIdx.x == 0 && blockIdx.x < 1024) a.trace[1024 + 4 * blockIdx.x] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  constexpr bool BFM = std::is_same<AT, ms_bf16m>::value;         // bf16 matrix arithmetic (needs R == 1)
  constexpr bool BFL = BFM;                                       // the bf16 LDS layout (8-byte channel-quad entries)
  constexpr bool WB = std::is_same<AT, ms_f32wb>::value || std::is_same<AT, ms_bf16wb>::value;        // Winograd form on independent 8x8-pixel blocks (WideGeoWB)
  constexpr int WFW = ms_wf_width<AT>::value;                                                          // (flat form: pixels per image row, else 0)
  constexpr bool WFL = WFW != 0;                                                                        // Winograd form on the flattened tile list of 20 / 24 / 28-pixel images (WideGeoWF)
  static_assert(!WFL || NT == 2, "flat form: two channel blocks per lane");
  constexpr bool WIN = std::is_same<AT, ms_f32w>::value || std::is_same<AT, ms_f32w32>::value || WB || WFL ||       // Winograd F(2x2, 3x3) (needs R == 1, NT <= 2, every chunk full)
                       std::is_same<AT, ms_bf16w>::value || std::is_same<AT, ms_bf16w32>::value;      // ... on bf16 storage
  constexpr int WTW = (std::is_same<AT, ms_f32w32>::value || std::is_same<AT, ms_bf16w32>::value) ? 32 : 64;
  constexpr int WHALVES = WTW / 32;                               // 32-pixel halves of a tile row = waves per tile row
  static_assert(!WIN || (R == 1 && NT <= 2 && AF), "Winograd mode: 4-row tiles, one or two channel blocks per lane, channel count a multiple of the chunk");
  using G = typename std::conditional<BFM, WideGeoBF<NT, PRO>, typename std::conditional<WB, WideGeoWB<(WB ? NT : 1), PRO>, typename std::conditional<WFL, WideGeoWF<(WFL ? NT : 1), PRO, (WFL ? WFW : 20)>,
            typename std::conditional<WIN, WideGeoW<(WIN ? NT : 1), PRO, WTW>, WideGeo<NT, PRO, R>>::type>::type>::type>::type;
  using IO = ActIO<AT>;
  constexpr int AB = IO::kBytes;
  static_assert(!BFM || R == 1, "bf16 MFMA mode: 4-row tiles");
  constexpr int TH = G::TH, TW = G::TW, CK = G::CK, IH = G::IH, RS = G::RS, PS = G::PS, WS = G::WS, BUF = G::BUF;
  constexpr int NQI = G::NQI, NHI = G::NHI, NWI = G::NWI, COUT_TILE = 16 * NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = __builtin_amdgcn_readfirstlane((int)(MS_TID >> 6)), lane = MS_TID & 63;      // wave-uniform by construction: keep it in a scalar register
  const bool producer = wave >= 4;
  const int ntiles = a.tiles_x * a.tiles_y, ncb = a.ncb;
  const int nitems = ((WB || WFL) ? 1 : a.N) * ntiles * ncb;          // (block form: a "tile" is a group of four blocks of the flattened (image, block row, block column) list; flat form: 64 tiles of the flattened tile list)
  // flat form: tiles per image, per batch; group g = tiles 64 g .. 64 g + 63 of the batch's list
  const int wf_T = (a.Hout >> 1) * (WFL ? G::TW / 2 : 1), wf_total = a.N * wf_T;
  const int nchunks = (a.cin_pad + CK - 1) / CK;
  // block form: block b of the flattened list -> (image, first row, first column); b beyond the list -> an empty block (rows beyond the image: nothing loaded or stored)
  const int wb_bx = (a.Wout + 7) >> 3, wb_by = (a.Hout + 7) >> 3, wb_total = a.N * wb_bx * wb_by;
  auto wb_decode = [&](int b, int& bn, int& y0, int& x0) {
    if (b >= wb_total) { bn = 0; y0 = a.Hout; x0 = 0; return false; }
    const int per = wb_bx * wb_by;
    bn = b / per; const int r = b - bn * per; const int byi = r / wb_bx;
    y0 = byi * 8; x0 = (r - byi * wb_bx) * 8;
    return true;
  };
  const int vb = ((int)gridDim.x % 8 == 0) ? (((int)blockIdx.x % 8) * ((int)gridDim.x / 8) + (int)blockIdx.x / 8) : (int)blockIdx.x;
  const int my_items = (nitems - vb + (int)gridDim.x - 1) / (int)gridDim.x;
  const int T = my_items * nchunks;
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto decode = [&](int it, int& n, int& tile, int& cb) { cb = it % ncb; const int t2 = it / ncb; tile = t2 % ntiles; n = t2 / ntiles; };

  // per-channel prologue coefficients are constant for the whole launch: ONE float4 {a, b, c, 0} per input channel in LDS behind the stage buffers, zero
  // beyond Cin.  The staging waves fetch them with one ds_read_b128 per item and chunk (was: three dependent global loads behind ~7 VALU instructions of
  // index arithmetic each - on a SIMD whose matrix pipe is saturated every VALU instruction of a staging wave waits behind an MFMA).
  float* cf_lds = smem + 2 * BUF;
  if constexpr (BFM) {                               // K groups 2, 3 (activation planes and weight entries) stay zero for the whole launch
    for (int i = MS_TID; i < 2 * BUF; i += 512) smem[i] = 0.f;
    __syncthreads();
  }
  // cross-workgroup finalize (`_xfin`, ms_conv_kernel.h): the table is filled by the MFMA waves from granules published inside this launch - in front of
  // barrier #0, while the staging waves already have their first chunk's global loads in flight (they read their coefficients behind that barrier)
  // (round 6: the table is ALWAYS filled by the MFMA waves in front of barrier #0 - from the coefficient arrays, or `_xfin` from the granules - while the staging waves
  //  have their first chunk's global loads in flight; until round 5 the array form was filled by all 512 threads behind a __syncthreads in FRONT of those loads:
  //  entry -> first MFMA 3.8 us (BatchNorm-apply prologue) / 4.8 us (two-tensor) against 2.9 us without a prologue, per-workgroup wall-clock stamps)
  const bool xf_pro = (PRO != 0) && (a.xf_tab != nullptr);

  if (producer) {
    // =========================================== PRODUCER waves ===========================================
#ifndef MS_WIDE_PRIO_SHIFT
#define MS_WIDE_PRIO_SHIFT 8
#endif
#ifndef MS_WIDE_STAGE_PRIO
#define MS_WIDE_STAGE_PRIO 3
#endif
    __builtin_amdgcn_s_setprio(MS_WIDE_STAGE_PRIO);
    const int tid = MS_TID - 256;
    const int sw_ = __builtin_amdgcn_readfirstlane(tid >> 6);      // staging wave 0..3 (block form: = the patch it stages)
    const int itid = WB ? (tid & 63) : tid;        // item numbering: per workgroup (256 staging threads), or per patch (64 lanes) in the block form
    constexpr int ISTR = WB ? 64 : 256;
    const int patch_off = WB ? sw_ * (IH * RS) : 0;
    // Winograd appendix (a.wu, MS_FETCH_WINO_U): the chunk's transformed weights come straight from HBM / L2 into the stage's weight region by LDS-DMA
    // (buffer_load_dwordx4 ... lds: 1 KB per wave-instruction, no vector register, no vector arithmetic) - 2 * NT instructions per staging wave instead of nine tap loads,
    // ~37 vector instructions and 16 LDS stores per thread.  Issued FIRST in the iteration that stores the chunk (the consumers left this buffer at the previous
    // barrier), covered by a counted vmcnt in front of this iteration's barrier: the next chunk's data loads, issued later, stay in flight.
    int wu_voff = 0;
    if constexpr (WIN) {
      const int pl = tid & 63;
      // lane -> (block j, channel-in-chunk, 4 output channels) inside one position's UP floats of the stage (see WideGeoW)
      if constexpr (NT == 2) wu_voff = 4 * ((pl >> 5) * nchunks * 2048 + ((pl & 31) >> 2) * 16 + (pl & 3) * 4);
      else wu_voff = 4 * ((pl >> 5) * 128 + ((pl & 31) >> 2) * 16 + (pl & 3) * 4);
    }
    auto dma_u = [&](float* buf, int cbi, int chunk_i) __attribute__((always_inline)) {
      if constexpr (WIN) {
        const ms_i32x4 ru = ms_dma_rsrc(a.wu);
        // the appendix holds ceil(Cout / 16) blocks: with two blocks per item and an odd count the last item's second block does not exist.  Its lanes (bit 5 of the
        // lane index) take an offset beyond the resource's range - the pieces read as zeros (the buffer range check looks at the VECTOR offset only) instead of whatever
        // lies behind the weights (round 6: a rare memory fault when the buffer ended its memory segment; Cout = 33 / 40 / 48 in the tests)
        const bool odd_tail = (NT == 2) && (2 * cbi + 1 >= ((a.Cout + 15) >> 4));
        const int u_voff = (odd_tail && (tid & 32)) ? (int)0x80000000 : wu_voff;
        const int sw = __builtin_amdgcn_readfirstlane(tid >> 6);
        const unsigned dst = ms_lds_addr(buf + CK * PS);
#pragma unroll
        for (int q = 0; q < 2 * NT; ++q) {
          const int piece = q * 4 + sw;                 // 256 floats of the weight region each
          const int so = (NT == 2) ? 4 * ((cbi * 2 * nchunks + chunk_i) * 2048 + piece * 128) : 4 * ((cbi * nchunks + chunk_i) * 2048 + piece * 256);
          ms_lds_dma16(ru, dst + 1024u * (unsigned)piece, u_voff, __builtin_amdgcn_readfirstlane(so));
        }
      }
    };
    int pb_n = 0;                                  // block form: the image of this wave's patch
    const int plane = a.Hs * a.Ws;                 // host checks Cin*plane < 2^31
    typedef unsigned mask_t;
    // every global address of the staging is (wave-uniform base of the chunk, biased back by one row + 4 so that no offset is negative) + a 32-bit BYTE offset
    // hoisted here: the loads take the scalar-base form (global_load ... v_off, s[base:base+1]) and cost no address arithmetic per chunk
    const int bias = a.Ws + 4;                     // elements; a multiple of 4: the 16-byte alignment of the quad loads is kept
    int q_lds[NQI], q_rc[NQI];                     // LDS offset (or -1), (row << 16) | (col_rel + 16)
    int h_lds[NHI], h_rc[NHI];
    unsigned q_off[NQI], h_off[NHI];               // byte offset of c*plane + (r-1)*W + col_rel + bias  (relative to the tile origin of channel c0)
    int q_cf[NQI], h_cf[NHI];                      // float4 index of the item's channel-in-chunk in the coefficient table
    static_assert(NQI <= 32 && NHI <= 32, "item masks are 32 bits");
    mask_t q_all = 0, h_all = 0;
#pragma unroll
    for (int j = 0; j < NQI; ++j) {
      const int it = itid + j * ISTR;
      q_lds[j] = -1; q_rc[j] = 0; q_off[j] = (unsigned)AB * (unsigned)bias; q_cf[j] = 0;
      if (it < G::Q_ITEMS) {
        const int f = it % (TW / 4), row = it / (TW / 4);
        const int r = row % IH, c = row / IH;
        if constexpr (BFL) q_lds[j] = (c << 20) | ((((c >> 2) * IH + r) * G::RSB + 4 * f + 1) * 8 + (c & 3) * 2);      // BYTE address of the entry's channel slot 
        else q_lds[j] = (c << 20) | (c * PS + patch_off + r * RS + 4 * f + 1);
        q_rc[j] = (r << 16) | (4 * f + 16);
        q_off[j] = (unsigned)AB * (unsigned)(c * plane + (r - 1) * a.Ws + 4 * f + bias);
        q_cf[j] = c;
        q_all |= 1u << j;
      }
    }
#pragma unroll
    for (int j = 0; j < NHI; ++j) {
      const int it = itid + j * ISTR;
      h_lds[j] = -1; h_rc[j] = 0; h_off[j] = (unsigned)AB * (unsigned)bias; h_cf[j] = 0;
      if (it < G::H_ITEMS) {
        const int h = it & 1, row = it >> 1;
        const int r = row % IH, c = row / IH;
        const int col_rel = h ? TW : -1;
        if constexpr (BFL) h_lds[j] = (c << 20) | ((((c >> 2) * IH + r) * G::RSB + col_rel + 1) * 8 + (c & 3) * 2);
        else h_lds[j] = (c << 20) | (c * PS + patch_off + r * RS + col_rel + 1);
        h_rc[j] = (r << 16) | (col_rel + 16);
        h_off[j] = (unsigned)AB * (unsigned)(c * plane + (r - 1) * a.Ws + col_rel + bias);
        h_cf[j] = c;
        h_all |= 1u << j;
      }
    }
    mask_t q_ok = 0, h_ok = 0;
    bool edge = false;
    int t_base = 0;
    unsigned q_offd[WFL ? NQI : 1];                // flat form: the item's own load offsets (band rows of the NEXT image sit a constant further)
    auto set_tile = [&](int tile) {
      if constexpr (WFL) {
        // band row r: first image n0 rows 2 ty0 - 1 + r for r < rows_a (the last of them is the row below the image), then image n0 + 1 rows r - rows_a - 1.
        // The hoisted offsets are linear in r from the band's first row; a row of the next image is (Cin * plane - (H + 2) * W) elements further, whatever the item.
        const int t0 = tile * 64, n0 = t0 / wf_T, l0 = t0 - n0 * wf_T, ty0 = l0 / (G::TW / 2);
        pb_n = n0;
        const int y0 = 2 * ty0;
        t_base = y0 * a.Ws;
        const int rows_a = a.Hin - y0 + 2;
        const bool live_b = n0 + 1 < a.N;
        const unsigned d_next = (unsigned)AB * (unsigned)(a.Cin * plane - (a.Hin + 2) * a.Ws);
        edge = true;
        q_ok = 0; h_ok = 0;
#pragma unroll
        for (int j = 0; j < NQI; ++j) {
          const int r = q_rc[j] >> 16;
          const bool part_b = r >= rows_a;
          const int y = part_b ? r - rows_a - 1 : y0 - 1 + r;
          const bool ok = (y >= 0) && (y < a.Hin) && (!part_b || live_b);
          q_ok |= (ok ? 1u : 0u) << j;
          q_offd[j] = q_off[j] + (part_b ? d_next : 0u);
        }
        q_ok &= q_all;
        return;
      }
      int y0, x0;
      bool live = true;
      if constexpr (WB) { live = wb_decode(tile * 4 + sw_, pb_n, y0, x0); }
      else { const int tx = tile % a.tiles_x, ty = tile / a.tiles_x; y0 = ty * TH; x0 = tx * TW; }
      t_base = live ? y0 * a.Ws + x0 : 0;              // (an empty block: every item masked, loads at the first pixel of image 0)
      const int ylo = 1 - y0, yhi = a.Hin - y0 + 1;               // ylo <= r < yhi
      const int xlo = 16 - x0, xhi = a.Win - x0 + 16;             // xlo <= (col_rel + 16) < xhi
      auto inside = [&](int rc) { const int r = rc >> 16, c = rc & 0xFFFF; return (r >= ylo) && (r < yhi) && (c >= xlo) && (c < xhi); };
      edge = (ylo > 0) || (yhi < IH) || (x0 + TW > a.Win) || !live;
      if (edge) {
        q_ok = 0;
#pragma unroll
        for (int j = 0; j < NQI; ++j) q_ok |= (inside(q_rc[j]) ? 1u : 0u) << j;
        q_ok &= q_all;
      } else {
        q_ok = q_all;
      }
      h_ok = 0;
#pragma unroll
      for (int j = 0; j < NHI; ++j) h_ok |= (inside(h_rc[j]) ? 1u : 0u) << j;
      h_ok &= h_all;
      if (!live) { q_ok = 0; h_ok = 0; }
    };
    float rq[BFM ? 1 : NQI][4], rq2[(PRO == 2 && !BFM) ? NQI : 1][4], rh[NHI], rh2[PRO == 2 ? NHI : 1];
    unsigned rqp[BFM ? NQI : 1][2], rqp2[(BFM && PRO == 2) ? NQI : 1][2];      // bf16-MFMA mode (16-channel chunks): the quads stay PACKED in registers until the LDS store
    float4 rw[NWI];
    float rww[WIN ? 9 : 1];                         // Winograd mode: the 9 taps of this thread's (input channel, output channel) pair
    mask_t l_q_ok = 0, l_h_ok = 0;                  // masks of the chunk held in registers
    bool l_edge = false, have_w = false;
    float ca[NQI], cb_[NQI], cc[PRO == 2 ? NQI : 1], hca[NHI], hcb[NHI], hcc[PRO == 2 ? NHI : 1];   // prologue coefficients of the chunk in registers

    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    auto load_data = [&](__amdgpu_buffer_rsrc_t r1, __amdgpu_buffer_rsrc_t r2, int soff, mask_t qm, mask_t hm, auto edge_tag) {
      constexpr bool EDGE = decltype(edge_tag)::value;
      const int origin = AB * bias;                     // masked items read the (valid) tile origin and are zeroed at the LDS store
#pragma unroll
      for (int j = 0; j < NQI; ++j) {
        int off = WFL ? (int)q_offd[WFL ? j : 0] : (int)q_off[j];
        if constexpr (EDGE) off = ((qm >> j) & 1u) ? off : origin;
        if constexpr (AB == 4) {
          const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r1, off, soff, 0);
          rq[j][0] = __uint_as_float(v.x); rq[j][1] = __uint_as_float(v.y); rq[j][2] = __uint_as_float(v.z); rq[j][3] = __uint_as_float(v.w);
          if constexpr (PRO == 2) {
            const u32x4_t u = __builtin_amdgcn_raw_buffer_load_b128(r2, off, soff, 0);
            rq2[j][0] = __uint_as_float(u.x); rq2[j][1] = __uint_as_float(u.y); rq2[j][2] = __uint_as_float(u.z); rq2[j][3] = __uint_as_float(u.w);
          }
        } else if constexpr (BFM) {
          const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r1, off, soff, 0);
          rqp[j][0] = v.x; rqp[j][1] = v.y;
          if constexpr (PRO == 2) { const u32x2_t u = __builtin_amdgcn_raw_buffer_load_b64(r2, off, soff, 0); rqp2[j][0] = u.x; rqp2[j][1] = u.y; }
        } else {                                      // bf16 storage: 4 values in 8 bytes, widened to fp32
          const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r1, off, soff, 0);
          rq[j][0] = __uint_as_float(v.x << 16); rq[j][1] = __uint_as_float(v.x & 0xFFFF0000u); rq[j][2] = __uint_as_float(v.y << 16); rq[j][3] = __uint_as_float(v.y & 0xFFFF0000u);
          if constexpr (PRO == 2) {
            const u32x2_t u = __builtin_amdgcn_raw_buffer_load_b64(r2, off, soff, 0);
            rq2[j][0] = __uint_as_float(u.x << 16); rq2[j][1] = __uint_as_float(u.x & 0xFFFF0000u); rq2[j][2] = __uint_as_float(u.y << 16); rq2[j][3] = __uint_as_float(u.y & 0xFFFF0000u);
          }
        }
      }
      // (flat form: the halo columns are never inside the image - NO loads, stated here and not left to dead-code elimination: the counted wait behind the weights'
      //  LDS-DMA below - kDataLoads - counts the loads this function issues)
#pragma unroll
      for (int j = 0; j < (WFL ? 0 : NHI); ++j) {
        const int off = ((hm >> j) & 1u) ? (int)h_off[j] : origin;       // the halo columns of the first / last tile of a row are outside the image
        if constexpr (AB == 4) {
          rh[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r1, off, soff, 0));
          if constexpr (PRO == 2) rh2[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r2, off, soff, 0));
        } else {
          rh[j] = __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r1, off, soff, 0) << 16);
          if constexpr (PRO == 2) rh2[j] = __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r2, off, soff, 0) << 16);
        }
      }
    };
    auto load_coefs = [&](int c0) {
      if constexpr (PRO != 0) {
        // the chunk's per-channel coefficients: one ds_read_b128 per item (zero beyond Cin)
        const float4* cf_c0 = reinterpret_cast<const float4*>(cf_lds) + c0;
#pragma unroll
        for (int j = 0; j < NQI; ++j) {
          const float4 cf = cf_c0[q_cf[j]];
          ca[j] = cf.x; cb_[j] = cf.y;
          if constexpr (PRO == 2) cc[j] = cf.z;
        }
#pragma unroll
        for (int j = 0; j < NHI; ++j) {
          const float4 cf = cf_c0[h_cf[j]];
          hca[j] = cf.x; hcb[j] = cf.y;
          if constexpr (PRO == 2) hcc[j] = cf.z;
        }
      }
    };
    auto load_chunk = [&](int n, int co0, int c0, bool load_w, bool coefs = true) {
      // buffer addressing: resource base = image n, biased back (scalar arithmetic); soffset = chunk + tile origin (scalar); voffset = the hoisted item offset:
      // `buffer_load_dwordx4 v, v_off, s[rsrc], s_off offen` - no vector address arithmetic per chunk (the host checks Cin*plane*4 < 2^31)
      if constexpr (WB || WFL) n = pb_n;                                                      // (block form: this wave's patch has its own image; flat form: the item's first image)
      const ptrdiff_t img_off = ((ptrdiff_t)n * a.Cin * plane - bias) * AB;                  // bytes
      char* img = const_cast<char*>(reinterpret_cast<const char*>(a.in)) + img_off;
      const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(img, 0, 0x7FFFFFFF, 0x00020000);
      char* img2 = (PRO == 2) ? const_cast<char*>(reinterpret_cast<const char*>(a.in2)) + img_off : img;
      const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(img2, 0, 0x7FFFFFFF, 0x00020000);
      const int soff = AB * (c0 * plane + t_base);
      const bool ragged = (c0 + CK > a.Cin);          // last chunk of a layer whose channel count is not a multiple of CK
      l_q_ok = q_ok; l_h_ok = h_ok; l_edge = edge || ragged || (a_dbg & 2);
      mask_t qm = q_ok, hm = h_ok;
      if (ragged) {
#pragma unroll
        for (int j = 0; j < NQI; ++j) if (c0 + (q_lds[j] >> 20) >= a.Cin) qm &= ~(1u << j);
#pragma unroll
        for (int j = 0; j < NHI; ++j) if (c0 + (h_lds[j] >> 20) >= a.Cin) hm &= ~(1u << j);
        l_q_ok = qm; l_h_ok = hm;
      }
      if (a_dbg & 2) { qm = 0; hm = 0; }                // timing-only: every lane reads the tile origin
      if (l_edge) load_data(r1, r2, soff, qm, hm, std::true_type{}); else load_data(r1, r2, soff, qm, hm, std::false_type{});
      if (coefs) load_coefs(c0);
      have_w = load_w;
      if constexpr (WIN) {
        if (load_w && tid < G::W_ITEMS && !a_has_wu) {
          {
            // buffer addressing: resource = the packed weights, scalar offset = (tap, first channel of the chunk, channel block), vector offset = the
            // thread's hoisted (channel-in-chunk, output channel) - nine loads, no vector address arithmetic (layers with many chunks re-stage per chunk)
            const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, 0x7FFFFFFF, 0x00020000);
            // thread -> (block j = tid >> 7, channel-in-chunk = (tid >> 4) & 7, output channel 16 j + (tid & 15)): a staging wave covers ONE block, so its LDS stores
            // below (position * UP + tid) are 64 consecutive dwords
            const int w_vo = 4 * (((tid >> 4) & 7) * a.cout_pad + 16 * (tid >> 7) + (tid & 15));
            const int tap_stride = 4 * a.cin_pad * a.cout_pad, w_so = 4 * (c0 * a.cout_pad + co0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) rww[tap] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rwt, w_vo, w_so + tap * tap_stride, 0));
          }
        }
      } else
      if (load_w) {
#pragma unroll
        for (int j = 0; j < NWI; ++j) {
          const int idx = tid + j * 256;
          const int j4 = idx % (COUT_TILE / 4);
          const int row = idx / (COUT_TILE / 4);      // tap*CK + c
          const int c = row % CK, tap = row / CK;
          rw[j] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (idx < G::W_ITEMS && c0 + c < a.cin_pad)
            rw[j] = *reinterpret_cast<const float4*>(a.w + ((size_t)tap * a.cin_pad + c0 + c) * a.cout_pad + co0 + j4 * 4);
        }
      }
    };

    auto store_chunk = [&](float* buf, auto edge_tag) {
      constexpr bool EDGE = decltype(edge_tag)::value;
      float* w_lds = buf + CK * PS;
#pragma unroll
      for (int j = 0; j < NQI; ++j) {
        const bool full = (j + 1) * ISTR <= G::Q_ITEMS;
        if (!full && q_lds[j] < 0) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v2 = 0.f;
          if constexpr (BFM) {
            v[e] = __uint_as_float((e & 1) ? (rqp[j][e >> 1] & 0xFFFF0000u) : (rqp[j][e >> 1] << 16));
            if constexpr (PRO == 2) v2 = __uint_as_float((e & 1) ? (rqp2[j][e >> 1] & 0xFFFF0000u) : (rqp2[j][e >> 1] << 16));
          } else {
            v[e] = rq[j][e];
            if constexpr (PRO == 2) v2 = rq2[j][e];
          }
          if constexpr (PRO == 1) v[e] = leaky(ca[j] * v[e] + cb_[j], a.slope);
          if constexpr (PRO == 2) v[e] = ca[j] * v[e] + (cb_[j] * v2 + cc[j]);
          if constexpr (EDGE) v[e] = ((l_q_ok >> j) & 1u) ? v[e] : 0.f;          // zero padding pads the tensor AFTER the prologue
        }
        if constexpr (BFM) {
          // round to bf16 and scatter into the channel slot of the four pixels' entries (8 bytes apart)
          char* dst = reinterpret_cast<char*>(buf) + (q_lds[j] & 0xFFFFF);
#pragma unroll
          for (int e = 0; e < 4; ++e) *reinterpret_cast<uint16_t*>(dst + 8 * e) = ms_to_bf16(v[e]);
        } else {
          float* dst = buf + (q_lds[j] & 0xFFFFF);
          dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];            // column 1 + 4f: dword stores (paired into ds_write2_b32)
        }
      }
#pragma unroll
      for (int j = 0; j < NHI; ++j) {
        if (h_lds[j] < 0) continue;
        if constexpr (WFL) { buf[h_lds[j] & 0xFFFFF] = 0.f; continue; }      // flat form: columns 0 / W + 1 of the band are the zero padding, every chunk
        float v = rh[j];
        if constexpr (PRO == 1) v = leaky(hca[j] * v + hcb[j], a.slope);
        if constexpr (PRO == 2) v = hca[j] * v + (hcb[j] * rh2[j] + hcc[j]);
        if constexpr (BFM) *reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(buf) + (h_lds[j] & 0xFFFFF)) = ms_to_bf16(((l_h_ok >> j) & 1u) ? v : 0.f);
        else buf[h_lds[j] & 0xFFFFF] = ((l_h_ok >> j) & 1u) ? v : 0.f;
      }
      if constexpr (WIN) {
        if (have_w && tid < G::W_ITEMS && !a_has_wu) {
          // U = G g G^T, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]: position p = 4*xi + nu (xi along ky, nu along kx)
          float t[4][3];
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float g0 = rww[kx], g1 = rww[3 + kx], g2 = rww[6 + kx];
            t[0][kx] = g0; t[1][kx] = 0.5f * ((g0 + g2) + g1); t[2][kx] = 0.5f * ((g0 + g2) - g1); t[3][kx] = g2;
          }
#pragma unroll
          for (int xi = 0; xi < 4; ++xi) {
            const float u0 = t[xi][0], u3 = t[xi][2];
            const float u1 = 0.5f * ((t[xi][0] + t[xi][2]) + t[xi][1]), u2 = 0.5f * ((t[xi][0] + t[xi][2]) - t[xi][1]);
            w_lds[(xi * 4 + 0) * G::UP + tid] = u0; w_lds[(xi * 4 + 1) * G::UP + tid] = u1;
            w_lds[(xi * 4 + 2) * G::UP + tid] = u2; w_lds[(xi * 4 + 3) * G::UP + tid] = u3;
          }
        }
      } else
      if (have_w) {
#pragma unroll
        for (int j = 0; j < NWI; ++j) {
          const int idx = tid + j * 256;
          if (idx < G::W_ITEMS) {
            const int j4 = idx % (COUT_TILE / 4);
            const int row = idx / (COUT_TILE / 4);
            if constexpr (BFM) {
              // entry (tap, g, cout) = channels 4g..4g+3 of one output channel: this item holds 4 output channels of ONE input channel
              const int c = row % CK, tap = row / CK;
              char* wb = reinterpret_cast<char*>(buf) + G::A_BYTES + ((tap * 4 + (c >> 2)) * COUT_TILE + j4 * 4) * 8 + (c & 3) * 2;
              const float wv[4] = {rw[j].x, rw[j].y, rw[j].z, rw[j].w};
#pragma unroll
              for (int e = 0; e < 4; ++e) *reinterpret_cast<uint16_t*>(wb + 8 * e) = ms_to_bf16(wv[e]);
            } else {
              *reinterpret_cast<float4*>(w_lds + row * WS + j4 * 4) = rw[j];
            }
          }
        }
      }
    };

    int key_cb[2] = {-1, -1}, key_c0[2] = {-1, -1};
    int item = vb, chunk = 0, n, tile, cb, tile_set = -1;
#ifdef MS_CONV_TRACE_BUILD
    const bool wtr = (a.trace != nullptr) && (MS_TID == 256) && (blockIdx.x < 1024);
    if (wtr) a.trace[1024 + 4096 + 1024 + 4 * blockIdx.x] = (long long)__builtin_amdgcn_s_memrealtime();          // hoisted set-up done
#endif
    decode(item, n, tile, cb);
    set_tile(tile); tile_set = tile;
    load_chunk(n, cb * COUT_TILE, 0, true, false);
#ifdef MS_CONV_TRACE_BUILD
    if (wtr) a.trace[1024 + 4096 + 1024 + 4 * blockIdx.x + 1] = (long long)__builtin_amdgcn_s_memrealtime();      // first chunk's loads issued
#endif
    lds_barrier();                                    // barrier #0 (matched by the consumers): the coefficient table is complete behind it
#ifdef MS_CONV_TRACE_BUILD
    if (wtr) a.trace[1024 + 4096 + 1024 + 4 * blockIdx.x + 2] = (long long)__builtin_amdgcn_s_memrealtime();      // behind barrier #0
#endif
    load_coefs(0);
#ifdef MS_CONV_TRACE_BUILD
    const bool tr = (a.trace != nullptr) && (blockIdx.x == 0) && (MS_TID == 256);
#else
    constexpr bool tr = false;
#endif
    // vector-memory loads load_chunk issues for one chunk's activations - EXACTLY: the counted wait behind the weights' LDS-DMA relies on it.  Round 6, found as a
    // run-to-run difference of ~1e-5 in one work item of a config-4 call: in the flat form the halo loads are dead (their mask is the constant 0), the compiler removed
    // them, and vmcnt(NQI + NHI) left the last NHI pieces of the DMA unwaited - staged weights read before they had landed, rarely, on a cold first call
    // (tools/check_dma_wait.py checks the built ISA: loads issued behind the DMA >= the count waited for)
    constexpr int kDataLoads = (PRO == 2 ? 2 : 1) * (NQI + (WFL ? 0 : NHI));
    for (int p = 0; p < T; ++p) {
      if (tr && p < 16) a.trace[128 + p * 4 + 0] = clock64();
      if (tr && p < 16) { a.trace[256 + p * 8 + 0] = clock64(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); a.trace[256 + p * 8 + 1] = clock64(); }      // (stamps: wait for the chunk's loads, separated from the arithmetic)
      const bool dma_now = WIN && (a_has_wu) && have_w;      // (have_w / cb / chunk describe chunk p here: they move on below)
      const int dma_cb = cb, dma_chunk = chunk;
      if (!(a_dbg & 8)) {
        if (l_edge) store_chunk(smem + (p & 1) * BUF, std::true_type{}); else store_chunk(smem + (p & 1) * BUF, std::false_type{});
      }
      if (tr && p < 16) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); a.trace[128 + p * 4 + 1] = clock64(); }
#ifdef MS_CONV_TRACE_BUILD
      if (wtr && p == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); a.trace[1024 + 4096 + 1024 + 4 * blockIdx.x + 3] = (long long)__builtin_amdgcn_s_memrealtime(); }      // first chunk stored
#endif
      key_cb[p & 1] = cb; key_c0[p & 1] = chunk * CK;
      // the weights of chunk p by LDS-DMA: behind this iteration's LDS stores (the consumers left the buffer at the previous barrier), in front of the next chunk's
      // data loads - so the compiler's own counted waits for THOSE never include a DMA piece, and the counted wait below leaves them in flight
      if constexpr (WIN) { if (dma_now) dma_u(smem + (p & 1) * BUF, dma_cb, dma_chunk); }
      if (tr && p < 16) a.trace[256 + p * 8 + 2] = clock64();
      if (p + 1 < T) {
        if (++chunk == nchunks) { chunk = 0; item += gridDim.x; decode(item, n, tile, cb); }
        if (tile != tile_set) { set_tile(tile); tile_set = tile; }
        if (tr && p < 16) a.trace[256 + p * 8 + 3] = clock64();
        const int b = (p + 1) & 1;
        load_chunk(n, cb * COUT_TILE, chunk * CK, !(key_cb[b] == cb && key_c0[b] == chunk * CK));
      }
      if (tr && p < 16) a.trace[128 + p * 4 + 2] = clock64();
      if constexpr (WIN) {
        if (dma_now) {                                  // the LDS-DMA of this chunk's weights has landed (loads complete in order; the younger data loads may still fly)
          if (p + 1 < T) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDataLoads) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      if (tr && p < 16) a.trace[256 + p * 8 + 4] = clock64();
      lds_barrier();                                  // barrier #(p+1): chunk p visible; consumers done with chunk p-1
      if (tr && p < 16) a.trace[128 + p * 4 + 3] = clock64();
    }
    return;
  }

  // =========================================== CONSUMER waves ===========================================
  // (no s_setprio here: the STAGING waves get the priority - measured 290.7 -> 295.2 steps/s against the opposite choice; a staging wave that
  //  loses issue arbitration to back-to-back MFMAs is what the MFMA waves end up waiting for at the barrier)
  const int m = lane & 15, k = lane >> 4;
#ifdef MS_CONV_TRACE_BUILD
  if (a.trace != nullptr && MS_TID == 0 && blockIdx.x < 1024) a.trace[1024 + 4096 + blockIdx.x] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
#endif
#ifdef MS_WIDE_PRIO_EXP
  // experiment (round 6): the two workgroups of a CU do not progress at the same rate (per-workgroup wall-clock stamps: the first-placed one finishes its 8 items in
  // ~41 us, the second in ~50 us, alone on the CU at the end).  1: static priority for the odd thread-group slot; 2: priority alternates between the slots in time windows
  const unsigned hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);
  const int tg_slot = (hw_id >> 16) & 1;
  if (MS_WIDE_PRIO_EXP == 1 && tg_slot) __builtin_amdgcn_s_setprio(1);
  auto prio_tick = [&]() {
    if (MS_WIDE_PRIO_EXP == 2) {
      const int win = (int)((__builtin_amdgcn_s_memrealtime() >> MS_WIDE_PRIO_SHIFT) & 1);
      if (win == tg_slot) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    }
  };
#else
  auto prio_tick = [&]() {};
#endif
  f32x4 acc[R][4][NT];                                  // [row of this wave][pixel-in-quad i][channel block j]: rows = lane-local pixel quads
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[r][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int a_lane = k * PS + (R * wave) * RS + 4 * m;  // window of staged row (R*wave + s): columns 4m .. 4m+5
  const int b_lane = CK * PS + k * WS + m;

  // one pipeline step = (4-channel group cg, staged row s of this wave's R+2): ONE window (ds_read_b128 + ds_read_b64) feeds output row r with kernel
  // row ky = s - r for every r it is valid for: 12*NT MFMAs for the first and last staged row, 24*NT in between (R = 2).  The B fragments of kernel
  // row ky are read once per channel group (3*NT ds_read_b32) and used by every output row.  Per accumulator the order of the K loop is (cg, ky, kx):
  // the same as with R = 1, so the two tile heights give bit-identical results.
  constexpr int NS = R + 2;
  auto load_win = [&](const float* buf, int cg, int st, float (&win)[6]) {
    const float* q = buf + a_lane + cg * 4 * PS + st * RS;
    const float4 v = *reinterpret_cast<const float4*>(q);
    const float2 w = *reinterpret_cast<const float2*>(q + 4);
    win[0] = v.x; win[1] = v.y; win[2] = v.z; win[3] = v.w; win[4] = w.x; win[5] = w.y;
  };
  auto load_b = [&](const float* buf, int cg, int ky, float (&bf)[3][NT]) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[kx][j] = buf[b_lane + ((ky * 3 + kx) * CK + cg * 4) * WS + j * 16];
  };
  // bf16-MFMA mode: one 16-channel chunk = for every staged row st (= kernel row ky) THREE ds_read_b128 (this lane's 6-pixel window of channel quad g = k:
  // 6 entries of 8 bytes) + 3*NT ds_read_b64 (weights of the three kernel columns) feed 12*NT v_mfma_f32_16x16x16_bf16 - the work of 48*NT fp32 MFMAs.
  typedef short bf4_t __attribute__((ext_vector_type(4)));
  typedef unsigned cu32x4_t __attribute__((ext_vector_type(4)));
  typedef unsigned cu32x2_t __attribute__((ext_vector_type(2)));
  auto compute_bf = [&](const float* buf, auto first_tag) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_tag)::value;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BFM) {
      const char* ab = reinterpret_cast<const char*>(buf) + ((k * IH + wave) * G::RSB + 4 * m) * 8;
      const char* bb = reinterpret_cast<const char*>(buf) + G::A_BYTES + (k * COUT_TILE + m) * 8;
      cu32x4_t wn[2][3];
      cu32x2_t bw[2][3][NT];
      auto ld = [&](int st, cu32x4_t (&w)[3], cu32x2_t (&b)[3][NT]) {
#pragma unroll
        for (int q = 0; q < 3; ++q) w[q] = *reinterpret_cast<const cu32x4_t*>(ab + st * G::RSB * 8 + 16 * q);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int j = 0; j < NT; ++j) b[kx][j] = *reinterpret_cast<const cu32x2_t*>(bb + ((st * 3 + kx) * 4 * COUT_TILE + 16 * j) * 8);
      };
      ld(0, wn[0], bw[0]);
#pragma unroll
      for (int st = 0; st < 3; ++st) {
        if (st + 1 < 3) ld(st + 1, wn[(st + 1) & 1], bw[(st + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned* wv = reinterpret_cast<const unsigned*>(&wn[st & 1][0]);          // 12 dwords: pixel p of the window = dwords 2p, 2p+1
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const cu32x2_t ap = {wv[2 * (i + kx)], wv[2 * (i + kx) + 1]};
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[0][i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(bf4_t, ap), __builtin_bit_cast(bf4_t, bw[st & 1][kx][j]),
                                                                        (FIRST && st == 0 && kx == 0) ? zero4 : acc[0][i][j], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // Winograd mode: wave = (tile row tr = wave >> 1: output rows 2tr, 2tr+1; half h = wave & 1: 32 pixels = 16 tiles of 2x2); MFMA M index = tile, N = output
  // channel, K = input channel.  Per 4-channel group: the lane's 4x4 input patch (tile m, channel 4cg+k) = 8 ds_read_b64, V = B^T d B in registers
  // (32 additions), 16 B fragments (one ds_read_b32 per position), 16 MFMAs - one per position, 16 independent accumulators.
  constexpr int WNT = WIN ? NT : 1;
  f32x4 accw[WNT][WIN ? 16 : 1];
  auto compute_w = [&](const float* buf, auto first_tag) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_tag)::value;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (WIN) {
      const float* ab = WB ? buf + k * PS + wave * (IH * RS) + (2 * (m >> 2)) * RS + 2 * (m & 3)
                           : buf + k * PS + (2 * (wave / WHALVES)) * RS + 32 * (wave % WHALVES) + 2 * m;
      const float* ub = buf + CK * PS + lane;
#pragma unroll
      for (int cg = 0; cg < CK / 4; ++cg) {
        // Register budget (128 with two workgroups per CU, NT = 1): 64 accumulators + up to 16 prefetched mask values leave ~30 for operands, so the order is fixed by
        // hand (sched_barrier): patch rows are read in the order the position rows need them (d0 d2 | d1 | d3), B fragments one position row ahead.
        // NT = 2 (one workgroup per CU, 256 registers): the SAME transformed patch row feeds the position row of both channel blocks - 8 MFMAs per 5 vector
        // instructions instead of 4.
        const float* pa = ab + cg * 4 * PS;
        const float* pu = ub + cg * 64;
        // the patch rows stay in the register PAIRS the 8-byte reads deliver: (d0,d1) and (d2,d3) of a row.  Row transform = 2 packed additions per
        // position row; column transform (a,b | c,e) -> (a-c, b+c, c-b, b-e) = one packed subtraction (v0, v3) + two scalar ones: 5 vector instructions
        // per 4 MFMAs, each computed one position row AHEAD of the MFMAs that read it (no wait states between a v_add and the MFMA behind it).
        typedef float f2 __attribute__((ext_vector_type(2)));
        auto ldrow = [&](int r, f2& lo, f2& hi) {
          lo = *reinterpret_cast<const f2*>(pa + r * RS);
          hi = *reinterpret_cast<const f2*>(pa + r * RS + 2);
        };
        auto ldu = [&](int xi, float (&u)[WNT][4]) {
#pragma unroll
          for (int j = 0; j < WNT; ++j)
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) u[j][nu] = pu[(xi * 4 + nu) * G::UP + j * 128];
        };
        auto vrow = [&](f2 ta, f2 tb, float (&v)[4]) {
          const f2 w = ta - tb;
          v[0] = w.x; v[3] = w.y; v[1] = ta.y + tb.x; v[2] = tb.x - ta.y;
        };
        auto mm = [&](int xi, const float (&v)[4], const float (&u)[WNT][4]) {
#pragma unroll
          for (int j = 0; j < WNT; ++j)
#pragma unroll
            for (int nu = 0; nu < 4; ++nu)
              accw[j][xi * 4 + nu] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[nu], u[j][nu], (FIRST && cg == 0) ? zero4 : accw[j][xi * 4 + nu], 0, 0, 0);
        };
        f2 d0a, d0b, d1a, d1b, d2a, d2b, d3a, d3b;
        float ua[WNT][4], ub2[WNT][4], va[4], vb[4];
        ldrow(0, d0a, d0b); ldrow(2, d2a, d2b); ldu(0, ua); ldrow(1, d1a, d1b); ldu(1, ub2);
        __builtin_amdgcn_sched_barrier(0);
        vrow(d0a - d2a, d0b - d2b, va);
        __builtin_amdgcn_sched_barrier(0);
        mm(0, va, ua); vrow(d1a + d2a, d1b + d2b, vb); ldrow(3, d3a, d3b);
        __builtin_amdgcn_sched_barrier(0);
        ldu(2, ua);
        mm(1, vb, ub2); vrow(d2a - d1a, d2b - d1b, va);
        __builtin_amdgcn_sched_barrier(0);
        ldu(3, ub2);
        mm(2, va, ua); vrow(d1a - d3a, d1b - d3b, vb);
        __builtin_amdgcn_sched_barrier(0);
        mm(3, vb, ub2);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // ---- Winograd mode, two channel blocks per lane (one workgroup per CU: ONE MFMA wave per SIMD, so nothing hides this wave's LDS latency but its own
  // schedule).  Software pipeline over the 4-channel groups, fixed register roles: patch rows d0..d3, B-fragment sets U[0..3] (one per position row), va / vb:
  //   S1: MM(0) [va, U0] | vb = V1(d1 + d2) | load d3, U2          S3: MM(2) [va, U2] | vb = V3(d1 - d3) | load d0', d2', U0' of the NEXT group
  //   S2: MM(1) [vb, U1] | va = V2(d2 - d1) | load U3              S4: MM(3) [vb, U3] | load d1', U1' | va = V0'(d0' - d2')
  // so every load has 8 MFMAs (256 cycles) between its issue and its first use.  The last group of a chunk passes the chunk barrier EARLY, between S2 and S3: by
  // then every LDS read of this chunk has been issued (lgkmcnt(0) in front of the barrier: and has returned), so the staging waves may overwrite the buffer, and
  // the next chunk's first operands - from the OTHER buffer, complete behind that barrier - are fetched under MM(2) / MM(3).  Same barrier count as the plain loop
  // (one per chunk); in the launch's very last chunk the barrier only meets the other MFMA waves (the staging waves have returned: s_barrier counts live waves)
  // and the prefetch reads stale, valid LDS that nobody uses.  Per accumulator the order of the K loop is the one-block kernel's: the same bits.
  typedef float w2f2 __attribute__((ext_vector_type(2)));
#ifdef MS_CONV_TRACE_BUILD
  const bool w2_tr = (a.trace != nullptr) && (blockIdx.x == 0) && (MS_TID == 0);
#else
  constexpr bool w2_tr = false;
#endif
  int w2_p = 0;                                        // (cycle stamps only) running chunk number
  w2f2 w2d[4][2];
  float w2u[4][WNT][4], w2va[4], w2vb[4];
  // flat form: the lane's patch offset depends on the item (tile -> band row, column): w2_aoff = the current item's, w2_aoff_pf = the one the cross-chunk prefetch of
  // the LAST chunk of an item uses (the next item's); every other form: one constant
  auto wf_aoff = [&](int grp) {
    const int t0 = grp * 64, n0 = t0 / wf_T, l0 = t0 - n0 * wf_T, ty0 = l0 / (G::TW / 2);
    const int t = min(t0 + 16 * wave + m, wf_total - 1);
    const int n = t / wf_T, l = t - n * wf_T, ty = l / (G::TW / 2), tx = l - ty * (G::TW / 2);
    const int top = (n == n0) ? 2 * (ty - ty0) : (a.Hin - 2 * ty0 + 2) + 2 * ty;      // band row of the patch's first row (image row 2 ty - 1)
    return k * PS + top * RS + 2 * tx;
  };
  int w2_aoff = WB ? k * PS + wave * (IH * RS) + (2 * (m >> 2)) * RS + 2 * (m & 3) : k * PS + (2 * (wave / WHALVES)) * RS + 32 * (wave % WHALVES) + 2 * m;
  int w2_aoff_pf = w2_aoff;
  auto w2_ldrow = [&](const float* pa, int r) __attribute__((always_inline)) {
    w2d[r][0] = *reinterpret_cast<const w2f2*>(pa + r * RS);
    w2d[r][1] = *reinterpret_cast<const w2f2*>(pa + r * RS + 2);
  };
  auto w2_ldu = [&](const float* pu, int xi) __attribute__((always_inline)) {
    if constexpr (WIN) {
#pragma unroll
      for (int j = 0; j < WNT; ++j)
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) w2u[xi][j][nu] = pu[(xi * 4 + nu) * G::UP + j * 128];
    }
  };
  auto w2_vrow = [&](w2f2 ta, w2f2 tb, float (&v)[4]) __attribute__((always_inline)) {
    const w2f2 w = ta - tb;
    v[0] = w.x; v[3] = w.y; v[1] = ta.y + tb.x; v[2] = tb.x - ta.y;
  };
  auto w2_pre_a = [&](const float* buf, int cg, int aoff) __attribute__((always_inline)) {
    const float* pa = buf + aoff + cg * 4 * PS;
    w2_ldrow(pa, 0); w2_ldrow(pa, 2); w2_ldu(buf + CK * PS + lane + cg * 64, 0);
  };
  auto w2_pre_b = [&](const float* buf, int cg, int aoff) __attribute__((always_inline)) {
    w2_ldrow(buf + aoff + cg * 4 * PS, 1); w2_ldu(buf + CK * PS + lane + cg * 64, 1);
    w2_vrow(w2d[0][0] - w2d[2][0], w2d[0][1] - w2d[2][1], w2va);
  };
  // aoff_n: the patch offset of the NEXT group's operands (the same item's, or - behind the early barrier of an item's last chunk, flat form - the next item's)
  auto w2_cg = [&](const float* buf, int cg, auto zero_tag, const float* nbuf, int ncg, auto barrier_tag, int aoff_n) __attribute__((always_inline)) {
    constexpr bool ZERO = decltype(zero_tag)::value, BAR = decltype(barrier_tag)::value;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (WIN) {
      const float* pa = buf + w2_aoff + cg * 4 * PS;
      const float* pu = buf + CK * PS + lane + cg * 64;
      auto mm = [&](int xi, const float (&v)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < WNT; ++j)
#pragma unroll
          for (int nu = 0; nu < 4; ++nu)
            accw[j][xi * 4 + nu] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[nu], w2u[xi][j][nu], ZERO ? zero4 : accw[j][xi * 4 + nu], 0, 0, 0);
      };
      // (loads FIRST in every group, pinned: the compiler otherwise sinks them behind the group's MFMAs and a load issued in front of the early barrier would
      //  be waited for with one MFMA of cover instead of eight)
      __builtin_amdgcn_sched_barrier(0);
      w2_ldrow(pa, 3); w2_ldu(pu, 2);
      __builtin_amdgcn_sched_barrier(0);
      mm(0, w2va); w2_vrow(w2d[1][0] + w2d[2][0], w2d[1][1] + w2d[2][1], w2vb);
      __builtin_amdgcn_sched_barrier(0);
      w2_ldu(pu, 3);
      __builtin_amdgcn_sched_barrier(0);
      mm(1, w2vb); w2_vrow(w2d[2][0] - w2d[1][0], w2d[2][1] - w2d[1][1], w2va);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (BAR) {
        if (w2_tr && w2_p < 16) a.trace[w2_p * 4 + 1] = clock64();
        lds_barrier();
        if (w2_tr && w2_p < 16) a.trace[w2_p * 4 + 2] = clock64();
        __builtin_amdgcn_sched_barrier(0);
      }
      w2_pre_a(nbuf, ncg, aoff_n);
      __builtin_amdgcn_sched_barrier(0);
      mm(2, w2va); w2_vrow(w2d[1][0] - w2d[3][0], w2d[1][1] - w2d[3][1], w2vb);
      __builtin_amdgcn_sched_barrier(0);
      w2_ldrow(nbuf + aoff_n + ncg * 4 * PS, 1); w2_ldu(nbuf + CK * PS + lane + ncg * 64, 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(3, w2vb); w2_vrow(w2d[0][0] - w2d[2][0], w2d[0][1] - w2d[2][1], w2va);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // one K-chunk (CK / 4 = 2 groups); p = the chunk's running number in this workgroup (selects the stage buffer)
  auto w2_chunk = [&](int p, auto first_tag) __attribute__((always_inline)) {
    const float* buf = smem + (p & 1) * BUF;
    const float* nbuf = smem + ((p + 1) & 1) * BUF;
    static_assert(!WIN || CK == 8, "two channel groups per chunk");
    w2_p = p;
    if (w2_tr && p < 16) a.trace[p * 4 + 0] = clock64();
    w2_cg(buf, 0, first_tag, buf, 1, std::false_type{}, w2_aoff);
    w2_cg(buf, 1, std::false_type{}, nbuf, 0, std::true_type{}, w2_aoff_pf);
    if (w2_tr && p < 16) a.trace[p * 4 + 3] = clock64();
  };

  // FULL = every channel group of the chunk is live: straight-line code; the guarded form only runs for a layer's ragged last chunk
  // FIRST = first K-chunk of an item: the first MFMA of every accumulator takes a zero C operand (no clearing pass after the epilogue)
  auto compute = [&](const float* buf, auto full_tag, int ncg, auto first_tag) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BFM) { compute_bf(buf, first_tag); return; }
    
    if constexpr (WIN) { compute_w(buf, first_tag); return; }
    float win[2][6], bfr[3][3][NT];
    load_win(buf, 0, 0, win[0]);
    load_b(buf, 0, 0, bfr[0]);
#pragma unroll
    for (int cg = 0; cg < CK / 4; ++cg) {
      if (FULL || cg < ncg) {
#pragma unroll
        for (int st = 0; st < NS; ++st) {
          const int t = cg * NS + st;
          if (st + 1 < NS) {
            load_win(buf, cg, st + 1, win[(t + 1) & 1]);
            if (st + 1 < 3) load_b(buf, cg, st + 1, bfr[st + 1]);
          } else if (cg + 1 < CK / 4 && (FULL || cg + 1 < ncg)) {
            load_win(buf, cg + 1, 0, win[(t + 1) & 1]);
            load_b(buf, cg + 1, 0, bfr[0]);                // kernel row 0 of this group was last used at staged row R - 1
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int ky = st - r;
            if (ky >= 0 && ky < 3) {
#pragma unroll
              for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                  for (int j = 0; j < NT; ++j)
                    acc[r][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(win[t & 1][i + kx], bfr[ky][kx][j],
                                                                        (FIRST && cg == 0 && ky == 0 && kx == 0) ? zero4 : acc[r][i][j], 0, 0, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };

  // ---- epilogue: a lane holds, per output row of its wave and channel block, 16 consecutive pixels (16k .. 16k+15) of tile row R*wave + rr for channel m ----
  float st_n = 0.f, st_mean[NT], st_m2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { st_mean[j] = 0.f; st_m2[j] = 0.f; }
  float bias_v[NT];
  float mk_sc[NT], mk_sh[NT], mk_mu[NT];              // epi_mode 3: forward BatchNorm map + mean of this lane's channels
  int bias_co0 = -1;
  auto load_bias = [&](int co0) {
    if (co0 == bias_co0) return;
    bias_co0 = co0;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int co = co0 + j * 16 + m;
      bias_v[j] = (a_has_bias && co < a.Cout) ? a.bias[co] : 0.f;
      if (a_epi == 3) {
        const float4 cf = (co < a.Cout) ? reinterpret_cast<const float4*>(a.mk_coef)[co] : make_float4(0.f, 0.f, 0.f, 0.f);
        mk_sc[j] = cf.x; mk_sh[j] = cf.y; mk_mu[j] = cf.z;
      }
    }
  };
  // epi_mode 3 with one channel block per lane: the 16 values of u this lane masks with (per output row) are requested BEFORE the MFMA loop of the
  // item's last K-chunk, so the epilogue does not wait for them (loading them inside the epilogue cost ~10 us per launch at 16->16 @16x256x256)
  constexpr bool UPRE = (NT == 1 && R == 1);
  float4 upre[UPRE ? R : 1][4];
  auto prefetch_u = [&](int n, int tile, int co0) __attribute__((always_inline)) {
    if constexpr (!UPRE) return;
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const int xb = tx * TW + 16 * k;
    const int co = co0 + m;
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      const int y = ty * TH + R * wave + rr;
      const int nvalid = (y < a.Hout && co < a.Cout) ? max(0, min(16, a.Wout - xb)) : 0;
      const size_t off = (((size_t)n * a.Cout + co) * a.Hout + y) * a.Wout + xb;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        upre[UPRE ? rr : 0][r] = (4 * r < nvalid) ? IO::ld4(a.mk_u, off + 4 * r) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto epilogue = [&](int n, int tile, int co0) __attribute__((always_inline)) {
    if (a_dbg & 16) {                                    // timing-only: no epilogue at all (upper bound of what hiding it can gain)
#pragma unroll
      for (int rr = 0; rr < R; ++rr)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[rr][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      return;
    }
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const int xb = tx * TW + 16 * k;
    // epi_mode 3 without the early prefetch (8-row tiles: no registers to park 2 x 16 values across the MFMA loop): request the u values of EVERY
    // row of this wave up front, so that their latency is paid once per item and overlaps the first row's arithmetic
    float4 ulate[(!UPRE && NT == 1) ? R : 1][4];
    if (!UPRE && NT == 1 && a_epi == 3) {
      const int co = co0 + m;
#pragma unroll
      for (int rr = 0; rr < R; ++rr) {
        const int y = ty * TH + R * wave + rr;
        const int nv = (y < a.Hout && co < a.Cout) ? max(0, min(16, a.Wout - xb)) : 0;
        const size_t off = (((size_t)n * a.Cout + co) * a.Hout + y) * a.Wout + xb;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          ulate[rr][r] = (4 * r < nv) ? IO::ld4(a.mk_u, off + 4 * r) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      const int y = ty * TH + R * wave + rr;
      const bool row_ok = y < a.Hout;
      int nvalid = 0;                                      // valid pixels among this lane's 16 (Wout % 4 == 0 on this path)
      if (row_ok) nvalid = max(0, min(16, a.Wout - xb));
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[rr][i][j][r] += bias_v[j];
      if (a_has_stats && nvalid > 0) {
        // PER-LANE running (count, mean, M2) of this lane's 16-pixel groups, Chan-merged group by group; the four lanes that share a channel are
        // merged once, at the end of the kernel.  (The first version reduced every tile across those lanes: six dependent ds_bpermute round trips
        // and two IEEE divisions per tile = ~2 k cycles of epilogue per item on an idle CU - tools/trace_conv.py with MS_CONV_DBG=15.)
        const float cnt = (float)nvalid;
        const float rc = (nvalid == 16) ? 0.0625f : __builtin_amdgcn_rcpf(cnt);
        const float nt_ = st_n + cnt;
        const float wgt = cnt * __builtin_amdgcn_rcpf(nt_);       // merge weight to 1 ulp: enters as d*wgt and d*d*n*wgt, both second-order terms
        const bool all16 = (nvalid == 16);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          float s0 = 0.f, s1 = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (all16 || 4 * r < nvalid) { s0 += acc[rr][0][j][r] + acc[rr][1][j][r]; s1 += acc[rr][2][j][r] + acc[rr][3][j][r]; }
          }
          const float mean = (s0 + s1) * rc;
          float q0 = 0.f, q1 = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (all16 || 4 * r < nvalid) {
              const float d0 = acc[rr][0][j][r] - mean, d1 = acc[rr][1][j][r] - mean, d2 = acc[rr][2][j][r] - mean, d3 = acc[rr][3][j][r] - mean;
              q0 += d0 * d0 + d1 * d1; q1 += d2 * d2 + d3 * d3;
            }
          }
          const float d = mean - st_mean[j];
          st_mean[j] += d * wgt;
          st_m2[j] += (q0 + q1) + d * d * st_n * wgt;
        }
        st_n = nt_;
      }
      if (a_epi == 3) {
        // g = acc * lrelu'(sc*u + sh); running sums of g and g*(u - mean) per channel in st_mean / st_m2 (act_bwd_reduce_kernel<1>)
        if (row_ok) {
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int co = co0 + j * 16 + m;
            if (co >= a.Cout) continue;
            const size_t off = (((size_t)n * a.Cout + co) * a.Hout + y) * a.Wout + xb;
            float4 uu[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (UPRE) uu[r] = upre[UPRE ? rr : 0][r];                   // fetched before the item's last K-chunk (prefetch_u)
              else if (NT == 1) uu[r] = ulate[rr][r];          // fetched at the top of this epilogue
              else if (4 * r < nvalid) uu[r] = IO::ld4(a.mk_u, off + 4 * r);
            }
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (4 * r < nvalid) {
                float4 v;
                v.x = acc[rr][0][j][r] * ((mk_sc[j] * uu[r].x + mk_sh[j] > 0.f) ? 1.f : a.mk_slope);
                v.y = acc[rr][1][j][r] * ((mk_sc[j] * uu[r].y + mk_sh[j] > 0.f) ? 1.f : a.mk_slope);
                v.z = acc[rr][2][j][r] * ((mk_sc[j] * uu[r].z + mk_sh[j] > 0.f) ? 1.f : a.mk_slope);
                v.w = acc[rr][3][j][r] * ((mk_sc[j] * uu[r].w + mk_sh[j] > 0.f) ? 1.f : a.mk_slope);
                IO::st4(a.out, off + 4 * r, v);
                s1 += (v.x + v.y) + (v.z + v.w);
                s2 += (v.x * (uu[r].x - mk_mu[j]) + v.y * (uu[r].y - mk_mu[j])) + (v.z * (uu[r].z - mk_mu[j]) + v.w * (uu[r].w - mk_mu[j]));
              }
            }
            st_mean[j] += s1; st_m2[j] += s2;
          }
        }
      } else if (row_ok && !(a_dbg & 4)) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int co = co0 + j * 16 + m;
          if (co >= a.Cout) continue;
          const size_t op = (((size_t)n * a.Cout + co) * a.Hout + y) * a.Wout + xb;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (4 * r < nvalid) {
              float4 v = make_float4(acc[rr][0][j][r], acc[rr][1][j][r], acc[rr][2][j][r], acc[rr][3][j][r]);
              if (a_epi == 1) { const float4 p = IO::ld4(a.out, op + 4 * r); v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
              IO::st4(a.out, op + 4 * r, v);
            }
          }
        }
      }
    }
  };

  // ---- Winograd mode: output transform Y = A^T M A (A^T = [[1,1,1,0],[0,1,-1,-1]]) in registers - a lane holds all 16 positions of its 4 tiles (4k..4k+3 of the
  // wave's 16) for channel m - then the epilogue on 2 rows x 8 consecutive pixels (x = 32h + 8k ..) per lane.  upre[0][2*row + quad] = the mask tensor's values.
  // block form: the MFMA wave's block of the current item (image, first row, first column; an empty block sits below the image: every row invalid) - set per item
  int wb_n = 0, wb_y0 = 0, wb_x0 = 0;
  bool wb_live = true;
  auto wb_set = [&](int tile) { if constexpr (WB) wb_live = wb_decode(tile * 4 + wave, wb_n, wb_y0, wb_x0); };
  auto wino_geo = [&](int tile, int& xb, int& y0) {
    if constexpr (WB) { xb = wb_x0; y0 = wb_y0 + 2 * k; }
    else { const int tx = tile % a.tiles_x, ty = tile / a.tiles_x; xb = tx * TW + 32 * (wave % WHALVES) + 8 * k; y0 = ty * TH + 2 * (wave / WHALVES); }
  };
  // interior tiles: buffer addressing as in epilogue_fast below - resource = image n, scalar offset = (channel block, row, first column of the wave's half),
  // vector offset = the lane's hoisted (channel m, pixel group k) offset; stores carry the row offset in the vector offset (see bstore4)
  const int w_plane = a.Hout * a.Wout;
  const int w_voff = WB ? AB * (m * w_plane + 2 * k * a.Wout) : AB * (m * w_plane + 8 * k);
  typedef unsigned wu32x4_t __attribute__((ext_vector_type(4)));
  auto w_rsrc = [&](const float* base, int n) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base)) + (ptrdiff_t)n * a.Cout * w_plane * AB, 0, 0x7FFFFFFF, 0x00020000);
  };
  auto w_soff = [&](int tile, int co0, int row) {
    if constexpr (WB) return AB * ((co0 * a.Hout + wb_y0 + row) * a.Wout + wb_x0);
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    return AB * ((co0 * a.Hout + ty * TH + 2 * (wave / WHALVES) + row) * a.Wout + tx * TW + 32 * (wave % WHALVES));
  };
  typedef unsigned wu32x2_t __attribute__((ext_vector_type(2)));
  constexpr int WQB = 4 * AB;                           // bytes of one quad
  auto w_load4 = [&](__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    if constexpr (AB == 4) {
      const wu32x4_t w = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
      return make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
    } else {
      const wu32x2_t w = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
      return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xFFFF0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xFFFF0000u));
    }
  };
  auto w_store4 = [&](__amdgpu_buffer_rsrc_t r, int voff, float4 v) {
    if constexpr (AB == 4) {
      const wu32x4_t w = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
      __builtin_amdgcn_raw_buffer_store_b128(w, r, voff, 0, 0);
    } else {
      const wu32x2_t w = {ms_pack_bf16x2(v.x, v.y), ms_pack_bf16x2(v.z, v.w)};
      __builtin_amdgcn_raw_buffer_store_b64(w, r, voff, 0, 0);
    }
  };
  // mask-tensor values of the item (epi_mode 3): 16 per lane.  With 64 accumulators there are no registers to park them across the MFMA loop, so they are
  // requested at the start of the item STRAIGHT INTO LDS (buffer_load_dwordx4 ... lds: lane l's 16 bytes land at base + 16 l, no vector register involved)
  // and read back by the epilogue: 4 KB per MFMA wave behind the coefficient table.
  float* u_lds = smem + 2 * BUF + (PRO != 0 ? 4 * (nchunks * CK) : 0) + wave * (1024 * WNT);      // (+ 1024 floats per channel block j)
  auto wino_interior = [&](int tile, int co0) {
    if constexpr (WB) return wb_live && (wb_x0 + 8 <= a.Wout) && (wb_y0 + 8 <= a.Hout) && (co0 + 16 <= a.Cout) && !(a_dbg & 32);
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    return (tx * TW + TW <= a.Wout) && (ty * TH + TH <= a.Hout) && (co0 + 16 <= a.Cout) && !(a_dbg & 32);
  };
  auto prefetch_u_w = [&](int n, int tile, int cob) __attribute__((always_inline)) {
    if constexpr (WIN) {
      // (LDS-DMA as inline assembly - ms_lds_dma16, ms_common.h: behind the compiler's builtin every later LDS access waited for vmcnt(0), i.e. for these loads)
      const ms_i32x4 ru = ms_dma_rsrc(reinterpret_cast<const char*>(a.mk_u) + (ptrdiff_t)n * a.Cout * w_plane * AB);
      const unsigned ul = ms_lds_addr(u_lds);
#pragma unroll
      for (int j = 0; j < WNT; ++j) {
        const int co0 = cob + 16 * j;
        if (!wino_interior(tile, co0)) continue;        // border tiles load their (masked) values inside the epilogue
#pragma unroll
        for (int row = 0; row < 2; ++row) {
          const int so = __builtin_amdgcn_readfirstlane(w_soff(tile, co0, row));
          if constexpr (AB == 4) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
              ms_lds_dma16(ru, ul + 4u * (unsigned)(j * 1024 + (2 * row + q) * 256), w_voff + 16 * q, so);
          } else {                                     // bf16 storage: the lane's 8 pixels of a row are ONE 16-byte transfer
            ms_lds_dma16(ru, ul + 4u * (unsigned)(j * 1024 + row * 256), w_voff, so);
          }
        }
      }
    }
  };
  // (the transform runs ONCE, before the epilogue variants branch: 64 accumulators must not stay live across their joins)
  auto wino_out = [&](int j, float (&o)[2][8]) __attribute__((always_inline)) {
    if constexpr (WIN) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s[2][4];
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
          const float m0 = accw[j][nu][r], m1 = accw[j][4 + nu][r], m2 = accw[j][8 + nu][r], m3 = accw[j][12 + nu][r];
          s[0][nu] = (m0 + m1) + m2; s[1][nu] = (m1 - m2) - m3;
        }
#pragma unroll
        for (int row = 0; row < 2; ++row) {
          o[row][2 * r] = (s[row][0] + s[row][1]) + s[row][2];
          o[row][2 * r + 1] = (s[row][1] - s[row][2]) - s[row][3];
        }
      }
    }
  };
  // j = the channel block of the lane's NT this call stores (co0 = its first channel); the running pixel count st_n is shared by the blocks of an item: every block
  // merges against the count BEFORE the item and leaves the new one in stn_next (the caller commits it after the last block)
  auto epilogue_w = [&](int n, int tile, int co0, int j, float (&o)[2][8], float& stn_next, auto interior_tag) __attribute__((always_inline)) {
    constexpr bool INT = decltype(interior_tag)::value;
    if constexpr (WIN) {
      if (a_dbg & 16) return;
      int xb, y0; wino_geo(tile, xb, y0);
      const int co = co0 + m;
      if (a_has_bias) {
#pragma unroll
        for (int row = 0; row < 2; ++row)
#pragma unroll
          for (int e = 0; e < 8; ++e) o[row][e] += bias_v[j];
      }
      const int nvx = INT ? 8 : max(0, min(8, a.Wout - xb));       // valid pixels among this lane's 8 per row (Wout % 4 == 0 on this path)
      bool ok[2][2];
#pragma unroll
      for (int row = 0; row < 2; ++row)
#pragma unroll
        for (int q = 0; q < 2; ++q) ok[row][q] = INT || ((y0 + row < a.Hout) && (4 * q < nvx));
      if (a_has_stats) {
        // per-lane running (count, mean, M2): this item's group = the valid quads of the lane's 2 x 8 pixels (see the generic epilogue)
        int cnt_i = 0; float sm = 0.f;
#pragma unroll
        for (int row = 0; row < 2; ++row)
#pragma unroll
          for (int q = 0; q < 2; ++q)
            if (ok[row][q]) { cnt_i += 4; sm += (o[row][4 * q] + o[row][4 * q + 1]) + (o[row][4 * q + 2] + o[row][4 * q + 3]); }
        if (INT || cnt_i > 0) {
          const float cnt = INT ? 16.f : (float)cnt_i;
          const float rc = (INT || cnt_i == 16) ? 0.0625f : __builtin_amdgcn_rcpf(cnt);
          const float mean = sm * rc;
          float qq = 0.f;
#pragma unroll
          for (int row = 0; row < 2; ++row)
#pragma unroll
            for (int q = 0; q < 2; ++q)
              if (ok[row][q]) {
                const float d0 = o[row][4 * q] - mean, d1 = o[row][4 * q + 1] - mean, d2 = o[row][4 * q + 2] - mean, d3 = o[row][4 * q + 3] - mean;
                qq += __builtin_fmaf(d1, d1, d0 * d0) + __builtin_fmaf(d3, d3, d2 * d2);      // (fused: 2 instructions less per quad, and closer to the exact sum)
              }
          const float nt_ = st_n + cnt;
          const float wgt = cnt * __builtin_amdgcn_rcpf(nt_);
          const float dd = mean - st_mean[j];
          st_mean[j] += dd * wgt;
          st_m2[j] += qq + dd * dd * st_n * wgt;
          stn_next = nt_;
        }
      }
      if (!INT && co >= a.Cout) return;
      const __amdgpu_buffer_rsrc_t ro = w_rsrc(a.out, n);
      if (a_epi == 3) {
        float4 um[2][2];
        if constexpr (INT) {
          // (the item's LDS-destined loads have landed: the caller waits ONCE per item, in front of the first block's epilogue)
          if constexpr (AB == 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) um[i >> 1][i & 1] = *reinterpret_cast<const float4*>(u_lds + j * 1024 + i * 256 + lane * 4);
          } else {
#pragma unroll
            for (int row = 0; row < 2; ++row) {
              const wu32x4_t w = *reinterpret_cast<const wu32x4_t*>(u_lds + j * 1024 + row * 256 + lane * 4);
              um[row][0] = make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xFFFF0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xFFFF0000u));
              um[row][1] = make_float4(__uint_as_float(w.z << 16), __uint_as_float(w.z & 0xFFFF0000u), __uint_as_float(w.w << 16), __uint_as_float(w.w & 0xFFFF0000u));
            }
          }
        } else {
#pragma unroll
          for (int row = 0; row < 2; ++row) {
            const size_t off = (((size_t)n * a.Cout + co) * a.Hout + (y0 + row)) * a.Wout + xb;
#pragma unroll
            for (int q = 0; q < 2; ++q) um[row][q] = ok[row][q] ? IO::ld4(a.mk_u, off + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int row = 0; row < 2; ++row) {
          const size_t off = (((size_t)n * a.Cout + co) * a.Hout + (y0 + row)) * a.Wout + xb;
          const int st_voff = w_voff + w_soff(tile, co0, row);
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            if (ok[row][q]) {
              const float4 uu = um[row][q];
              float4 v;
              v.x = o[row][4 * q] * ((mk_sc[j] * uu.x + mk_sh[j] > 0.f) ? 1.f : a.mk_slope);
              v.y = o[row][4 * q + 1] * ((mk_sc[j] * uu.y + mk_sh[j] > 0.f) ? 1.f : a.mk_slope);
              v.z = o[row][4 * q + 2] * ((mk_sc[j] * uu.z + mk_sh[j] > 0.f) ? 1.f : a.mk_slope);
              v.w = o[row][4 * q + 3] * ((mk_sc[j] * uu.w + mk_sh[j] > 0.f) ? 1.f : a.mk_slope);
              if constexpr (INT) w_store4(ro, st_voff + WQB * q, v); else IO::st4(a.out, off + 4 * q, v);
              s1 += (v.x + v.y) + (v.z + v.w);
              s2 += __builtin_fmaf(v.y, uu.y - mk_mu[j], v.x * (uu.x - mk_mu[j])) + __builtin_fmaf(v.w, uu.w - mk_mu[j], v.z * (uu.z - mk_mu[j]));
            }
          }
        }
        st_mean[j] += s1; st_m2[j] += s2;
      } else if (a_epi == 1) {
        float4 pv[2][2];
#pragma unroll
        for (int row = 0; row < 2; ++row) {
          const size_t off = (((size_t)n * a.Cout + co) * a.Hout + (y0 + row)) * a.Wout + xb;
          const int so = w_soff(tile, co0, row);
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            if constexpr (INT) pv[row][q] = w_load4(ro, w_voff + WQB * q, so);
            else pv[row][q] = ok[row][q] ? IO::ld4(a.out, off + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
#pragma unroll
        for (int row = 0; row < 2; ++row) {
          const size_t off = (((size_t)n * a.Cout + co) * a.Hout + (y0 + row)) * a.Wout + xb;
          const int st_voff = w_voff + w_soff(tile, co0, row);
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            if (ok[row][q]) {
              const float4 v = make_float4(o[row][4 * q] + pv[row][q].x, o[row][4 * q + 1] + pv[row][q].y, o[row][4 * q + 2] + pv[row][q].z, o[row][4 * q + 3] + pv[row][q].w);
              if constexpr (INT) w_store4(ro, st_voff + WQB * q, v); else IO::st4(a.out, off + 4 * q, v);
            }
          }
        }
      } else if (a_epi == 6) {
        // 2x2-POOLED store (MS_EPI_POOL2): out is [N, Cout, Hout/2, Wout/2] and receives the sum of every 2x2 block of the result - the lane's 2 rows x 8 pixels
        // are four such blocks (the Winograd output tiles themselves), summed in ms_pool2_sum's order (a.x + a.y) + (b.x + b.y) over the values as they would
        // have been STORED (bf16 storage: rounded first): the same bits as ms_conv2d + ms_pool2_sum, a quarter of the bytes written and none read back.
        auto rt = [](float v) { if constexpr (AB == 2) return IO::up(ms_to_bf16(v)); else return v; };
        const size_t offp = (((size_t)n * a.Cout + co) * (size_t)(a.Hout >> 1) + (size_t)(y0 >> 1)) * (size_t)(a.Wout >> 1) + (size_t)(xb >> 1);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          if (ok[0][q] && ok[1][q]) {
            const float p0 = (rt(o[0][4 * q]) + rt(o[0][4 * q + 1])) + (rt(o[1][4 * q]) + rt(o[1][4 * q + 1]));
            const float p1 = (rt(o[0][4 * q + 2]) + rt(o[0][4 * q + 3])) + (rt(o[1][4 * q + 2]) + rt(o[1][4 * q + 3]));
            IO::st2(a.out, offp + 2 * q, make_float2(p0, p1));
          }
        }
      } else if (!(a_dbg & 4)) {
#pragma unroll
        for (int row = 0; row < 2; ++row) {
          const size_t off = (((size_t)n * a.Cout + co) * a.Hout + (y0 + row)) * a.Wout + xb;
          const int st_voff = w_voff + w_soff(tile, co0, row);
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            if (ok[row][q]) {
              const float4 v = make_float4(o[row][4 * q], o[row][4 * q + 1], o[row][4 * q + 2], o[row][4 * q + 3]);
              if constexpr (INT) w_store4(ro, st_voff + WQB * q, v); else IO::st4(a.out, off + 4 * q, v);
            }
          }
        }
      }
    }
  };

  // ---- flat form: the lane holds four CONSECUTIVE tiles of the batch's tile list (64 grp + 16 wave + 4 k + r, r = 0..3: o[row][2 r], o[row][2 r + 1]) for channel co0 + m -
  // each its own (image, tile row, tile column): 8-byte loads / stores tile by tile.  Same epilogue arithmetic as epilogue_w; the statistics / activation-backward sums
  // group the lane's values differently (per tile list, not per image row): the tables agree with the tiled form's to summation order, `out` bit for bit.
  int wf_po[4];                                         // element offset of (image, channel 0, row 2 ty, column 2 tx) of the lane's four tiles; -1: beyond the list
  auto wf_set = [&](int grp) {
    if constexpr (WFL) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = grp * 64 + 16 * wave + 4 * k + r;
        const int n = t / wf_T, l = t - n * wf_T, ty = l / (G::TW / 2), tx = l - ty * (G::TW / 2);
        wf_po[r] = (t < wf_total) ? (n * a.Cout * a.Hout + 2 * ty) * a.Wout + 2 * tx : -1;      // (the host checks N * Cout * H * W < 2^31)
      }
    }
  };
  auto epilogue_wf = [&](int co0, int j, float (&o)[2][8], float& stn_next) __attribute__((always_inline)) {
    if constexpr (WFL) {
      const int co = co0 + m;
      if (a_has_bias) {
#pragma unroll
        for (int row = 0; row < 2; ++row)
#pragma unroll
          for (int e = 0; e < 8; ++e) o[row][e] += bias_v[j];
      }
      bool tv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) tv[r] = wf_po[r] >= 0;
      if (a_has_stats) {
        int cnt_i = 0; float sm = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (tv[r]) { cnt_i += 4; sm += (o[0][2 * r] + o[0][2 * r + 1]) + (o[1][2 * r] + o[1][2 * r + 1]); }
        if (cnt_i > 0) {
          const float cnt = (float)cnt_i;
          const float rc = (cnt_i == 16) ? 0.0625f : __builtin_amdgcn_rcpf(cnt);
          const float mean = sm * rc;
          float qq = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (tv[r]) {
              const float d0 = o[0][2 * r] - mean, d1 = o[0][2 * r + 1] - mean, d2 = o[1][2 * r] - mean, d3 = o[1][2 * r + 1] - mean;
              qq += __builtin_fmaf(d1, d1, d0 * d0) + __builtin_fmaf(d3, d3, d2 * d2);
            }
          const float nt_ = st_n + cnt;
          const float wgt = cnt * __builtin_amdgcn_rcpf(nt_);
          const float dd = mean - st_mean[j];
          st_mean[j] += dd * wgt;
          st_m2[j] += qq + dd * dd * st_n * wgt;
          stn_next = nt_;
        }
      }
      if (co >= a.Cout) return;
      const int cpl = co * a.Hout * a.Wout;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (!tv[r]) continue;
#pragma unroll
        for (int row = 0; row < 2; ++row) {
          const size_t off = (size_t)(wf_po[r] + cpl + row * a.Wout);
          float2 v = make_float2(o[row][2 * r], o[row][2 * r + 1]);
          if (a_epi == 3) {
            const float2 uu = IO::ld2(a.mk_u, off);
            v.x *= ((mk_sc[j] * uu.x + mk_sh[j] > 0.f) ? 1.f : a.mk_slope);
            v.y *= ((mk_sc[j] * uu.y + mk_sh[j] > 0.f) ? 1.f : a.mk_slope);
            s1 += v.x + v.y;
            s2 += __builtin_fmaf(v.y, uu.y - mk_mu[j], v.x * (uu.x - mk_mu[j]));
          } else if (a_epi == 1) {
            const float2 pv = IO::ld2(a.out, off);
            v.x += pv.x; v.y += pv.y;
          }
          if (!(a_dbg & 4)) IO::st2(a.out, off, v);
        }
      }
      if (a_epi == 3) { st_mean[j] += s1; st_m2[j] += s2; }
    }
  };

  // ---- interior items (every pixel, row and channel of the tile exists; 4-row tiles): no masks, and buffer addressing - resource base = image n,
  // scalar offset = (channel block, row, tile column), vector offset = the lane's hoisted (channel m, pixel group k) offset: the 64-bit address arithmetic
  // of the generic epilogue (~15 vector instructions per row) and its per-quad selects are gone.  Same arithmetic, same order: bit-identical results.
  const int out_plane = a.Hout * a.Wout;
  const int ep_voff = AB * (m * out_plane + 16 * k);
  constexpr int QB = 4 * AB;                          // bytes of one quad (4 consecutive pixels)
  typedef unsigned eu32x4_t __attribute__((ext_vector_type(4)));
  typedef unsigned eu32x2_t __attribute__((ext_vector_type(2)));
  auto img_rsrc = [&](const float* base, int n) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base)) + (ptrdiff_t)n * a.Cout * out_plane * AB, 0, 0x7FFFFFFF, 0x00020000);
  };
  auto bload4 = [&](__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    if constexpr (AB == 4) {
      const eu32x4_t w = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
      return make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
    } else {
      const eu32x2_t w = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
      return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xFFFF0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xFFFF0000u));
    }
  };
  // STORES take the row offset in the VECTOR offset (one v_add per row), never in the scalar-offset field.  Measured on MI355X: a 16-byte buffer store with a
  // register soffset whose data registers are overwritten by the next vector instruction stores the NEW value in lanes 12-15 of every 16-lane row of the
  // second data dword (16 -> 16 @2x64x64: channels 12-15, pixel 13 of every 16 wrong).  LLVM's hazard recogniser only pads ">64-bit store data overwritten by
  // a VALU write" when soffset is NOT a register (GCNHazardRecognizer::createsVALUHazard), so with a constant soffset the compiler inserts the wait state itself.
  auto bstore4 = [&](__amdgpu_buffer_rsrc_t r, int voff, float4 v) {
    if constexpr (AB == 4) {
      const eu32x4_t w = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
      __builtin_amdgcn_raw_buffer_store_b128(w, r, voff, 0, 0);
    } else {
      const eu32x2_t w = {ms_pack_bf16x2(v.x, v.y), ms_pack_bf16x2(v.z, v.w)};
      __builtin_amdgcn_raw_buffer_store_b64(w, r, voff, 0, 0);
    }
  };
  auto tile_interior = [&](int tile, int co0) {
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    return (R == 1) && (tx * TW + TW <= a.Wout) && (ty * TH + TH <= a.Hout) && (co0 + COUT_TILE <= a.Cout);
  };
  auto row_soff = [&](int tile, int co) {               // byte offset of (channel co, this wave's row, first column of the tile) inside one image
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    return AB * ((co * a.Hout + ty * TH + wave) * a.Wout + tx * TW);
  };
  auto prefetch_u_fast = [&](int n, int tile, int co0) __attribute__((always_inline)) {
    if constexpr (!UPRE) return;
    const __amdgpu_buffer_rsrc_t ru = img_rsrc(a.mk_u, n);
    const int so = row_soff(tile, co0);
#pragma unroll
    for (int r = 0; r < 4; ++r) upre[0][r] = bload4(ru, ep_voff + QB * r, so);
  };
  auto epilogue_fast = [&](int n, int tile, int co0) __attribute__((always_inline)) {
    if (a_dbg & 16) return;                              // timing-only: no epilogue at all
    const __amdgpu_buffer_rsrc_t ro = img_rsrc(a.out, n);
    if (a_has_bias) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[0][i][j][r] += bias_v[j];
    }
    if (a_has_stats) {
      const float nt_ = st_n + 16.f;
      const float wgt = 16.f * __builtin_amdgcn_rcpf(nt_);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { s0 += acc[0][0][j][r] + acc[0][1][j][r]; s1 += acc[0][2][j][r] + acc[0][3][j][r]; }
        const float mean = (s0 + s1) * 0.0625f;
        float q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d0 = acc[0][0][j][r] - mean, d1 = acc[0][1][j][r] - mean, d2 = acc[0][2][j][r] - mean, d3 = acc[0][3][j][r] - mean;
          q0 += d0 * d0 + d1 * d1; q1 += d2 * d2 + d3 * d3;
        }
        const float d = mean - st_mean[j];
        st_mean[j] += d * wgt;
        st_m2[j] += (q0 + q1) + d * d * st_n * wgt;
      }
      st_n = nt_;
    }
    if (a_epi == 3) {
      const __amdgpu_buffer_rsrc_t ru = img_rsrc(a.mk_u, n);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int so = row_soff(tile, co0 + 16 * j);
        const int st_voff = ep_voff + so;
        float4 uu[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) uu[r] = UPRE ? upre[0][r] : bload4(ru, ep_voff + QB * r, so);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float4 v;
          v.x = acc[0][0][j][r] * ((mk_sc[j] * uu[r].x + mk_sh[j] > 0.f) ? 1.f : a.mk_slope);
          v.y = acc[0][1][j][r] * ((mk_sc[j] * uu[r].y + mk_sh[j] > 0.f) ? 1.f : a.mk_slope);
          v.z = acc[0][2][j][r] * ((mk_sc[j] * uu[r].z + mk_sh[j] > 0.f) ? 1.f : a.mk_slope);
          v.w = acc[0][3][j][r] * ((mk_sc[j] * uu[r].w + mk_sh[j] > 0.f) ? 1.f : a.mk_slope);
          bstore4(ro, st_voff + QB * r, v);
          s1 += (v.x + v.y) + (v.z + v.w);
          s2 += (v.x * (uu[r].x - mk_mu[j]) + v.y * (uu[r].y - mk_mu[j])) + (v.z * (uu[r].z - mk_mu[j]) + v.w * (uu[r].w - mk_mu[j]));
        }
        st_mean[j] += s1; st_m2[j] += s2;
      }
    } else if (a_epi == 1) {
      // accumulate into the tensor: its own block - merged with the plain store the compiler guards the adds with selects and a vmcnt wait that, on
      // gfx9, also waits for the STORES of the previous item (measured: +3 us per launch on the plain epilogue)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int so = row_soff(tile, co0 + 16 * j);
        const int st_voff = ep_voff + so;
        float4 prev[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) prev[r] = bload4(ro, ep_voff + QB * r, so);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          bstore4(ro, st_voff + QB * r, make_float4(acc[0][0][j][r] + prev[r].x, acc[0][1][j][r] + prev[r].y, acc[0][2][j][r] + prev[r].z, acc[0][3][j][r] + prev[r].w));
      }
    } else if (!(a_dbg & 4)) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int st_voff = ep_voff + row_soff(tile, co0 + 16 * j);
#pragma unroll
        for (int r = 0; r < 4; ++r) bstore4(ro, st_voff + QB * r, make_float4(acc[0][0][j][r], acc[0][1][j][r], acc[0][2][j][r], acc[0][3][j][r]));
      }
    }
  };

  int item = vb, n, tile, cb;
  decode(item, n, tile, cb);
  if constexpr (WB) { wb_set(tile); n = wb_n; }         // (block form: this wave's own block and image)
  load_bias(cb * COUT_TILE);
  if constexpr (PRO != 0) {
    if (xf_pro) {
      unsigned xf_tag; int xf_nparts;
      xfin_header(a, xf_tag, xf_nparts);
      if (!xfin_produce(a, xf_tag, xf_nparts)) __builtin_amdgcn_s_sleep(30);
      xfin_fill<(PRO == 2 ? 3 : 2)>(a, cf_lds, nchunks * CK, xf_tag, vb & (kXfinRep - 1), MS_TID, 256, 0.f, 0.f);
    } else {
      for (int c = MS_TID; c < nchunks * CK; c += 256) {
        float4 cf = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < a.Cin) { cf.x = a.pro_a[c * a.pro_cstride]; cf.y = a.pro_b[c * a.pro_cstride]; if constexpr (PRO == 2) cf.z = a.pro_c[c * a.pro_cstride]; }
        reinterpret_cast<float4*>(cf_lds)[c] = cf;
      }
    }
  }
  lds_barrier();                                      // barrier #0
  lds_barrier();                                      // barrier #1: chunk 0 is in buffer 0
#ifdef MS_CONV_TRACE_BUILD
  const bool tr = (a.trace != nullptr) && (blockIdx.x == 0) && (MS_TID == 0);
  if (tr) { a.trace[500] = clock64(); a.trace[501] = (long long)__builtin_amdgcn_s_memrealtime(); }     // shader clock vs the 100 MHz constant clock
  if (a.trace != nullptr && threadIdx.x == 0 && blockIdx.x < 1024) a.trace[1024 + 4 * blockIdx.x + 1] = (long long)__builtin_amdgcn_s_memrealtime();
#else
  constexpr bool tr = false;
#endif
  // (the L2-prefetch experiment of round 2 - MFMA lanes touching the cold second tensor two chunks ahead, MS_CONV_PF - measured slower and was removed:
  //  profiles/r02_experiments.txt section 6)
  // Item loop with the first K-chunk peeled (its MFMAs start from a zero C operand).  With AF the guarded MFMA loop is not instantiated, so the
  // accumulators keep one register assignment across the whole loop (no copies where the variants used to join).
  constexpr bool W2 = WIN && NT == 2;
  if constexpr (WFL) { w2_aoff = wf_aoff(tile); w2_aoff_pf = w2_aoff; }
  if constexpr (W2) { w2_pre_a(smem, 0, w2_aoff); w2_pre_b(smem, 0, w2_aoff); }      // the first group's operands of chunk 0
  {
    int p = 0;
    auto mfma_chunk = [&](int ch, auto first_tag) __attribute__((always_inline)) {
      prio_tick();
      if constexpr (W2) { w2_chunk(p, first_tag); return; }
      else {
      const int ncg = AF ? CK / 4 : min(CK / 4, (a.cin_pad - ch * CK) / 4);
      if (tr && p < 16) a.trace[p * 4 + 0] = clock64();
      if (a_dbg & 1) {
        if constexpr (WIN) { if (decltype(first_tag)::value) { for (int j = 0; j < WNT; ++j) for (int q = 0; q < 16; ++q) accw[j][q] = f32x4{0.f, 0.f, 0.f, 0.f}; } }
        if (decltype(first_tag)::value) {
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < NT; ++j) acc[r][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      } else if (AF) {
        compute(smem + (p & 1) * BUF, std::true_type{}, ncg, first_tag);
      } else {
        compute(smem + (p & 1) * BUF, std::false_type{}, ncg, first_tag);
      }
      if (tr && p < 16) a.trace[p * 4 + 1] = clock64();
      }
    };
    for (int it = 0; it < my_items; ++it) {
      const int co0 = cb * COUT_TILE;
      constexpr bool FAST = (NT == 1 && R == 1 && !WIN);      // (two channel blocks per lane: the fast epilogue's extra live values push the kernel past 128 registers = one workgroup per CU)
      const bool interior = FAST && !(a_dbg & 32) && tile_interior(tile, co0);      // (dbg bit 32: generic epilogue everywhere - A/B switch)
      auto pre_u = [&]() __attribute__((always_inline)) {
        if constexpr (FAST) { if (interior && !(a_dbg & 64)) { prefetch_u_fast(n, tile, co0); return; } }
        if constexpr (WIN) { prefetch_u_w(n, tile, co0); return; }
        else prefetch_u(n, tile, co0);
      };
      // direct form: at the START of the item (the registers are reserved anyway, and in a step the mask tensor is cold).  Winograd form (LDS-DMA): BEHIND the first
      // chunk - the accumulator set-up of an item waits for vmcnt(0) (the previous item's stores still hold their data registers), and a prefetch issued in front of
      // that wait is simply waited for: its HBM latency was exposed once per item until round 4 (found in the ISA; the remaining chunks now cover it)
      if (UPRE && !WIN && a_epi == 3) pre_u();
      // flat form: the cross-chunk prefetch inside an item's LAST chunk fetches the NEXT item's first operands - at that item's patch offsets
      auto wf_next = [&]() __attribute__((always_inline)) {
        if constexpr (WFL) {
          if (it + 1 < my_items) { int nn, nt, ncb_; decode(item + (int)gridDim.x, nn, nt, ncb_); w2_aoff_pf = wf_aoff(nt); }
          else w2_aoff_pf = w2_aoff;
        }
      };
      if constexpr (WFL) { wf_set(tile); if (nchunks == 1) wf_next(); }
      mfma_chunk(0, std::true_type{});
      if (WIN && !WFL && a_epi == 3) pre_u();
      for (int ch = 1; ch < nchunks; ++ch) {
        if constexpr (!W2) lds_barrier();              // (two-block Winograd form: the chunk barrier sits inside the previous chunk - w2_cg)
        if (!W2 && tr && p < 16) a.trace[p * 4 + 3] = clock64();
        ++p;
        if constexpr (WFL) { if (ch == nchunks - 1) wf_next(); }
        mfma_chunk(ch, std::false_type{});
      }
      bool done = false;
      if constexpr (FAST) { if (interior) { epilogue_fast(n, tile, co0); done = true; } }
      if constexpr (WIN) {
        if (a_epi == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the item's LDS-destined mask values have landed (once, not per block: vmcnt also counts the previous block's stores)
        float stn_next = st_n;
#pragma unroll
        for (int j = 0; j < WNT; ++j) {
          float o[2][8];
          wino_out(j, o);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (WFL) epilogue_wf(co0 + 16 * j, j, o, stn_next);
          else if (wino_interior(tile, co0 + 16 * j)) epilogue_w(n, tile, co0 + 16 * j, j, o, stn_next, std::true_type{}); else epilogue_w(n, tile, co0 + 16 * j, j, o, stn_next, std::false_type{});
        }
        st_n = stn_next;
        if constexpr (WFL) w2_aoff = w2_aoff_pf;
        done = true;
      }
      if (!done) epilogue(n, tile, co0);
      if (!W2 && tr && p < 16) a.trace[p * 4 + 2] = clock64();
      item += gridDim.x;
      if (it + 1 < my_items) {
        decode(item, n, tile, cb); load_bias(cb * COUT_TILE);
        if constexpr (WB) { wb_set(tile); n = wb_n; }
        if constexpr (!W2) lds_barrier();
        if (!W2 && tr && p < 16) a.trace[p * 4 + 3] = clock64();
      }
      ++p;
    }
  }
#ifdef MS_CONV_TRACE_BUILD
  if (tr) { a.trace[502] = clock64(); a.trace[503] = (long long)__builtin_amdgcn_s_memrealtime(); }
  if (a.trace != nullptr && threadIdx.x == 0 && blockIdx.x < 1024) a.trace[1024 + 4 * blockIdx.x + 2] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  if (a_has_stats) conv_table_tail<NT, true>(a, smem, vb, ncb, st_n, st_mean, st_m2);
  else if (a_epi == 3) conv_table_tail<NT, false>(a, smem, vb, ncb, 0.f, st_mean, st_m2);
#ifdef MS_CONV_TRACE_BUILD
  if (a.trace != nullptr && threadIdx.x == 0 && blockIdx.x < 1024) a.trace[1024 + 4 * blockIdx.x + 3] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
}

template <int NT, int PRO, int R, bool AF, typename AT, int FX = -1>
int launch_conv_wide_t(ConvArgs a, hipStream_t st) {
  using G = typename std::conditional<std::is_same<AT, ms_bf16m>::value, WideGeoBF<NT, PRO>,
                                      typename std::conditional<std::is_same<AT, ms_f32w>::value || std::is_same<AT, ms_bf16w>::value, WideGeoW<(NT <= 2 ? NT : 1), PRO, 64>,
                                      typename std::conditional<std::is_same<AT, ms_f32w32>::value || std::is_same<AT, ms_bf16w32>::value, WideGeoW<(NT <= 2 ? NT : 1), PRO, 32>,
                                      typename std::conditional<std::is_same<AT, ms_f32wb>::value || std::is_same<AT, ms_bf16wb>::value, WideGeoWB<(NT <= 2 ? NT : 1), PRO>,
                                      typename std::conditional<(ms_wf_width<AT>::value != 0), WideGeoWF<(NT <= 2 ? NT : 1), PRO, (ms_wf_width<AT>::value != 0 ? ms_wf_width<AT>::value : 20)>,
                                                                WideGeo<NT, PRO, R>>::type>::type>::type>::type>::type;
  const size_t cin_tab = (PRO != 0) ? (size_t)cdiv(a.cin_pad, G::CK) * G::CK : 0;              // coefficient table: one float4 per input channel of the padded chunks
  const size_t lds_bytes = sizeof(float) * (2 * (size_t)G::BUF + 4 * cin_tab) + ((std::is_same<AT, ms_f32w>::value || std::is_same<AT, ms_f32w32>::value || std::is_same<AT, ms_bf16w>::value || std::is_same<AT, ms_bf16w32>::value ||
                                                                                      std::is_same<AT, ms_f32wb>::value || std::is_same<AT, ms_bf16wb>::value) ? NT * 16 * 1024 : 0);      // (flat form: no landing zone - its mask values are read tile by tile in the epilogue)      // Winograd mode: + the mask tensor's landing zone (4 KB per MFMA wave and channel block)
  if (lds_bytes > 160 * 1024) { set_error("ms_conv2d: %d input channels exceed the LDS coefficient table", a.Cin); return MS_ERR_INVALID; }
  static std::once_flag attr_once;                     // one flag per instantiation; the attribute itself is immutable afterwards
  std::call_once(attr_once, []() { (void)hipFuncSetAttribute((const void*)conv_wide_kernel<NT, PRO, R, AF, AT, FX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)); });
  constexpr bool kWB = std::is_same<AT, ms_f32wb>::value || std::is_same<AT, ms_bf16wb>::value;
  a.tiles_x = cdiv(a.Wout, G::TW); a.tiles_y = cdiv(a.Hout, G::TH);
  if (kWB) { a.tiles_x = cdiv(a.N * a.tiles_x * a.tiles_y, 4); a.tiles_y = 1; }      // block form: groups of four blocks of the flattened (image, block row, block column) list
  constexpr bool kWF = ms_wf_width<AT>::value != 0;
  if (kWF) { a.tiles_x = cdiv(a.N * (a.Hout / 2) * (a.Wout / 2), 64); a.tiles_y = 1; }      // flat form: groups of 64 tiles of the batch's tile list
  a.ncb = cdiv(a.Cout, 16 * NT);
  const long nitems = (long)((kWB || kWF) ? 1 : a.N) * a.tiles_x * a.tiles_y * a.ncb;
  int per_cu = std::max(1, std::min(2, (int)((160 * 1024) / (lds_bytes + 256))));      // workgroups per CU of the persistent grid
  per_cu = std::max(1, std::min(per_cu, conv_resident_per_cu((const void*)conv_wide_kernel<NT, PRO, R, AF, AT, FX>, lds_bytes)));      // (co-residency of the whole grid: ms_conv_kernel.h)
  long nblocks = std::min<long>(nitems, (long)num_cus() * per_cu);
  if (nblocks > a.ncb) nblocks -= nblocks % a.ncb;
  MS_LAUNCH((conv_wide_kernel<NT, PRO, R, AF, AT, FX>), dim3((unsigned)nblocks), dim3(512), lds_bytes, st, a);
  return check_launch("conv_wide");
}
// the "fixed launch facts" code of a call (conv_wide_kernel's FX), for dispatchers that instantiate the hot combinations: epilogue kind | statistics | bias | appendix
inline int conv_wide_fx(const ConvArgs& a) {
  if (a.dbg != 0 || a.epi_mode < 0 || a.epi_mode > 7) return -1;
  return a.epi_mode | (a.stats != nullptr ? kFxStats : 0) | (a.bias != nullptr ? kFxBias : 0) | (a.wu != nullptr ? kFxWu : 0);
}
// Winograd form: the combinations of (prologue, epilogue kind, statistics, bias) the engines launch are instantiated with compile-time launch facts (fp32 storage, weights
// from the appendix); anything else - and every call under an ablation bit - takes the generic instantiation.  Forward convs: bias + statistics, plain store.  Data-gradients:
// no bias, no statistics; plain / accumulate / activation-backward / 2x2-pooled store.
template <int NT, int PRO, typename WT>
int launch_wino_fx(const ConvArgs& a, hipStream_t st) {
  constexpr bool F32 = std::is_same<WT, ms_f32w>::value || std::is_same<WT, ms_f32w32>::value || std::is_same<WT, ms_f32wb>::value || ms_wf_width<WT>::value != 0;
  if constexpr (F32) {
    const int fx = conv_wide_fx(a);
    constexpr int FWD = 0 | kFxStats | kFxBias | kFxWu;
    if constexpr (PRO != 2) { if (fx == FWD) return launch_conv_wide_t<NT, PRO, 1, true, WT, FWD>(a, st); }
    if constexpr (PRO != 1) {
      if (fx == (0 | kFxWu)) return launch_conv_wide_t<NT, PRO, 1, true, WT, (0 | kFxWu)>(a, st);
      if (fx == (1 | kFxWu)) return launch_conv_wide_t<NT, PRO, 1, true, WT, (1 | kFxWu)>(a, st);
    }
    if constexpr (PRO == 2) {
      if (fx == (3 | kFxWu)) return launch_conv_wide_t<NT, PRO, 1, true, WT, (3 | kFxWu)>(a, st);
      if (fx == (6 | kFxWu)) return launch_conv_wide_t<NT, PRO, 1, true, WT, (6 | kFxWu)>(a, st);
    }
  }
  return launch_conv_wide_t<NT, PRO, 1, true, WT>(a, st);
}
template <int NT, int PRO, int R, bool AF>
int launch_conv_wide_af(const ConvArgs& a, hipStream_t st) {
  if (a.act_bf16) {
    if constexpr (R == 1) {
      if (a.act_bf16 == 2) {                         // bf16 matrix arithmetic: 16-channel chunks, zero-filled beyond Cin - one instantiation per (NT, PRO)
        if constexpr (AF) return launch_conv_wide_t<NT, PRO, 1, true, ms_bf16m>(a, st);
        else return launch_conv_wide_t<NT, PRO, 1, true, ms_bf16m>(a, st);
      }
      return launch_conv_wide_t<NT, PRO, R, AF, ms_bf16>(a, st);         // (the 8-row experiment variant is fp32 only)
    } else { set_error("ms_conv2d_bf16: 8-row tiles are built for fp32 storage only"); return MS_ERR_INVALID; }
  }
  return launch_conv_wide_t<NT, PRO, R, AF, float>(a, st);
}
template <int NT, int PRO, int R>
int launch_conv_wide_r(const ConvArgs& a, hipStream_t st) {
  return (a.cin_pad % WideGeo<NT, PRO, R>::CK == 0) ? launch_conv_wide_af<NT, PRO, R, true>(a, st) : launch_conv_wide_af<NT, PRO, R, false>(a, st);
}

// tile height: 4 rows.  The 8-row variant (R = 2: two output rows per MFMA wave, bit-identical results) is kept behind the option "conv.wide_rows" = 8: in isolation it is
// 2-3 % faster on the activation-backward data-gradient at 16->16 @16x256x256 (72.8 vs 74.7 us), in the step it is slower (333.5 vs 336.3 steps/s, twice each).
inline int conv_wide_rows(const ConvArgs& a, int nt) {
  const int force = opt(OPT_CONV_WIDE_ROWS);
  if (force != 8 || nt != 1) return 4;     // (two channel blocks per lane x two rows per wave need 188 registers: one workgroup per CU - not built for it)
  const long items8 = (long)a.N * cdiv(a.Wout, 64) * cdiv(a.Hout, 8) * cdiv(a.Cout, 16 * nt);
  return (items8 >= 4L * num_cus()) ? 8 : 4;
}

// wide-read path: 3x3 stride 1, plain fetch, 16-byte aligned rows, per-channel prologue coefficients; implemented in ms_conv_inst_w.hip
bool conv_wide_eligible(const ConvArgs& a, int ks, int stride, int fetch, bool vec);
int conv_dispatch_wide(const ConvArgs& a, int nt, hipStream_t st);
bool conv_wide_is_wino(const ConvArgs& a);      // conv_dispatch_wide would run the Winograd form (the only one with the pooled epilogue, epi_mode 6)
int conv_dispatch_wide8(const ConvArgs& a, int nt, hipStream_t st);      // ms_conv_inst_w2.hip: the 8-row tiles

}  // namespace ms
