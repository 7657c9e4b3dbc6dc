// Host side of the fused inference BasicBlock launch (include/hrp.h, hrp_block_*); kernel: conv_block.h.
#include "conv_block.h"
#include <cstring>

namespace hrp {

// the channel count when the fused kernel takes the block, else 0
static int block_channels(const hrp_block_desc& b) {
  const hrp_conv_desc &a = b.conv1, &c = b.conv2;
  hrp_conv_desc p1 = a, p2 = c;
  // geometry / layout of both convolutions as the row-strip kernel wants it (conv2 reads the intermediate: same geometry as x)
  p1.y = (void*)a.x;
  p2.x = a.x;
  const int C = row_channels(p1);
  if ((C != 32 && C != 64) || row_channels(p2) != C) return 0;
  if (a.N != c.N || a.H != c.H || a.W != c.W || a.H % 4) return 0;
  if (!a.scale || !a.shift || !c.scale || !c.shift || a.relu != 1 || c.relu != 1) return 0;
  if (a.res || a.stats || c.stats || a.pro_mode || c.pro_mode || a.bnb_x || c.bnb_x || c.res_mask) return 0;
  if (!c.res || c.res != a.x || c.res_pitch != C || !c.y || !a.w || !c.w) return 0;
  if (((uintptr_t)a.scale | (uintptr_t)a.shift | (uintptr_t)c.scale | (uintptr_t)c.shift) % 16) return 0;
  return C;
}

template <int C0, int C1>
static int block_launch_t(const BlkArgs& A, int grid, int lds, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)block_kernel<C0, C1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((block_kernel<C0, C1>), dim3(grid), dim3(512), lds, s, A);
  return check_launch("block_kernel");
}

}  // namespace hrp

using namespace hrp;

extern "C" int hrp_block_channels(const hrp_block_desc* d) {
  static const bool off = false;
  return (d && !off) ? block_channels(*d) : 0;
}

extern "C" int64_t hrp_block_table_bytes(void) { return (int64_t)sizeof(BlkArgs); }

extern "C" int hrp_block_prepare(const hrp_block_desc* descs, int n, void* table, hrp_block_info* info) {
  HRP_REQUIRE(descs && info, "block: null pointer");
  HRP_REQUIRE(n >= 1 && n <= HRP_BLOCK_MAX, "block: n=%d is outside 1..%d", n, HRP_BLOCK_MAX);
  memset(info, 0, sizeof(*info));
  HRP_REQUIRE(n == 1 || (block_channels(descs[0]) == 32 && block_channels(descs[1]) == 64),
              "block: two problems of one launch are a 32-channel and a 64-channel block, in this order");
  // bands per image: one workgroup per image once the batch fills the chip with the launches that run next to each other
  // (B = 64: 2 problems x 64 images per trunk); small batches are cut into bands of >= 8 rows
  static const int target = 64;
  int grid = 0, lds = 0;
  BlkArgs A;
  memset(&A, 0, sizeof(A));
  for (int i = 0; i < n; ++i) {
    const int C = block_channels(descs[i]);
    HRP_REQUIRE(C != 0, "block: problem %d is not a 32- / 64-channel inference BasicBlock (scale / shift / relu on both convolutions, "
                "conv2.res == conv1.x)", i);
    {
      hrp_conv_desc p1 = descs[i].conv1, p2 = descs[i].conv2;       // (the intermediate has no address: checked with x's)
      p1.y = (void*)p1.x;
      p2.x = p1.x;
      int rc = conv_check(&p1);
      if (rc == HRP_OK) rc = conv_check(&p2);
      if (rc != HRP_OK) return rc;
    }
    const hrp_conv_desc &a = descs[i].conv1, &c = descs[i].conv2;
    int bands = 1;
    while (a.N * bands * 2 <= target && a.H % (bands * 2 * 4) == 0 && a.H / (bands * 2) >= 8) bands *= 2;
    info->first_wg[i] = grid;
    info->bands[i] = bands;
    grid += a.N * bands;
    const int l = C == 32 ? BlkCfg<32>::LDS_BYTES : BlkCfg<64>::LDS_BYTES;
    lds = l > lds ? l : lds;
    BlkProblem& q = A.q[i];
    q.x = a.x; q.y = c.y; q.w1 = a.w; q.w2 = c.w;
    q.sc1 = a.scale; q.sh1 = a.shift; q.sc2 = c.scale; q.sh2 = c.shift;
    q.N = a.N; q.H = a.H; q.bands = bands; q.band_rows = a.H / bands;
    q.w1_ntaps = a.w_ntaps; q.w2_ntaps = c.w_ntaps;
    for (int t = 0; t < 9; ++t) {
      q.wslot1[(a.dy[t] + 1) * 3 + a.dx[t] + 1] = a.wtap[t];
      q.wslot2[(c.dy[t] + 1) * 3 + c.dx[t] + 1] = c.wtap[t];
    }
    q.fd_bands = make_fastdiv(bands);
    A.first_wg[i] = info->first_wg[i];
    A.C[i] = C;
  }
  A.n = n; A.nwg = grid;
  info->n = n; info->grid = grid; info->lds_bytes = lds;
  if (table) *(BlkArgs*)table = A;
  return HRP_OK;
}

extern "C" int hrp_block_launch(const void* table, const hrp_block_info* info, void* stream) {
  HRP_REQUIRE(table && info && info->n >= 1 && info->n <= HRP_BLOCK_MAX && info->grid > 0, "block launch: bad arguments");
  const BlkArgs& A = *(const BlkArgs*)table;
  HRP_REQUIRE(A.n == info->n && A.nwg == info->grid, "block launch: table and info do not belong together");
  hipStream_t s = (hipStream_t)stream;
  const int c0 = A.C[0], c1 = A.n == 2 ? A.C[1] : 0;
  if (c0 == 32 && c1 == 64) return block_launch_t<32, 64>(A, info->grid, info->lds_bytes, s);
  if (c0 == 32 && c1 == 0) return block_launch_t<32, 0>(A, info->grid, info->lds_bytes, s);
  if (c0 == 64 && c1 == 0) return block_launch_t<64, 0>(A, info->grid, info->lds_bytes, s);
  set_error("block launch: channel combination (%d, %d)", c0, c1);
  return HRP_ERR_ARG;
}

#ifdef HRP_TIMELINE
extern "C" int hrp_debug_block_timeline(void* dst, int clear) {
  if (dst) (void)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_block_timeline), sizeof(unsigned long long) * 256 * 2 * 8);
  if (clear) {
    void* p = nullptr;
    (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_block_timeline));
    (void)hipMemset(p, 0, sizeof(unsigned long long) * 256 * 2 * 8);
  }
  return 0;
}
#endif
