// bf16 instantiations of the tile convolution (conv_tile.h) and the C entry point hrp_conv2d_fwd.
// (The fp32 kernels live in conv_fwd_f32.hip, the batched kernels in conv_batch_{bf16,f32}.hip: four translation
// units, for build time only.)
#include "conv_row.h"
#include "conv_pw.h"

namespace hrp {

int launch_conv_f32(const hrp_conv_desc& d, hipStream_t s);
int launch_conv_f32x3(const hrp_conv_desc& d, hipStream_t s);

int conv_check(const hrp_conv_desc* d) {
  HRP_REQUIRE(d && d->x && d->w && d->y, "conv: null pointer");
  HRP_REQUIRE(d->ntaps >= 1 && d->ntaps <= HRP_MAX_TAPS, "conv: ntaps=%d", d->ntaps);
  HRP_REQUIRE(d->dtype == HRP_F32 || d->dtype == HRP_BF16 || d->dtype == HRP_F32X3, "conv: dtype=%d", d->dtype);
  const int vec = d->dtype == HRP_BF16 ? 8 : 4;
  HRP_REQUIRE(d->Cin % vec == 0 && d->x_pitch % vec == 0 && (uintptr_t)d->x % 16 == 0,
              "conv: input channels / pitch must be multiples of %d elements (Cin=%d pitch=%d)", vec, d->Cin, d->x_pitch);
  HRP_REQUIRE((uintptr_t)d->w % 16 == 0, "conv: packed weights must be 16-byte aligned");
  HRP_REQUIRE(d->w_cout_pad % 32 == 0 && d->w_cout_pad >= d->Cout, "conv: w_cout_pad=%d Cout=%d", d->w_cout_pad, d->Cout);
  HRP_REQUIRE(d->N > 0 && d->Ho > 0 && d->Wo > 0 && d->Cout > 0 && d->Cin > 0, "conv: empty problem");
  HRP_REQUIRE(d->in_stride >= 1 && d->out_stride >= 1, "conv: strides");
  HRP_REQUIRE((d->scale == nullptr) == (d->shift == nullptr), "conv: scale and shift go together");
  PwPlan pwp;
  const bool lean_kernel = hrp_conv_rowstrip_channels(d) != 0 || pw_plan(*d, pwp) != 0;      // (their own eligibility checks cover bnb_*)
  if (d->bnb_x && d->bnb_mask && !lean_kernel) {   // BatchNorm-backward reduce in the epilogue of the general tile program: only the plain vector store path computes it
    const int sz = d->dtype == HRP_BF16 ? 2 : 4;
    HRP_REQUIRE(d->stats && d->bnb_mask && d->bnb_consts, "conv: bnb_x needs stats, bnb_mask and bnb_consts");
    HRP_REQUIRE(!d->res && !d->relu && !d->bias && !d->scale, "conv: bnb_x excludes res / relu / bias / scale");
    HRP_REQUIRE(d->out_stride == 1 && d->y_H == d->Ho && d->y_W == d->Wo, "conv: bnb_x needs a launch that covers y");
    HRP_REQUIRE(d->Cout % vec == 0 && d->bnb_mask_pitch >= d->Cout / vec, "conv: bnb_x needs whole vectors (Cout=%d)", d->Cout);
    HRP_REQUIRE((uintptr_t)d->y % 16 == 0 && ((size_t)d->y_pitch * sz) % 16 == 0 && (uintptr_t)d->bnb_x % 16 == 0 &&
                ((size_t)d->bnb_x_pitch * sz) % 16 == 0 && (uintptr_t)d->bnb_consts % 16 == 0, "conv: bnb_x alignment");
  }
  // input transforms / the mask-less epilogue reduce exist in the row-strip kernel only (conv_row.h)
  if (d->pro_mode != 0 || d->pro_side || d->pro_side2 || d->pro_mask || d->res_mask)
    HRP_REQUIRE(hrp_conv_rowstrip_channels(d) != 0, "conv: pro_mode / pro_side / pro_mask / res_mask need a row-strip problem (hrp_conv_rowstrip_channels)");
  if (d->tail_mode) HRP_REQUIRE(pw_plan(*d, pwp) != 0, "conv: tail_mode needs a pointwise problem with Cin 32 / 64 and Cout %% 64 == 0 (hrp_conv_pointwise)");
  if (d->bnb_x && (!d->bnb_mask || d->res))
    HRP_REQUIRE(lean_kernel, "conv: a mask-less bnb_x / bnb_x with res needs a row-strip or pointwise problem (hrp_conv_rowstrip_channels, hrp_conv_pointwise)");
  return HRP_OK;
}

}  // namespace hrp

extern "C" int hrp_conv_rowstrip_channels(const hrp_conv_desc* d) {
  return d ? hrp::row_channels(*d) : 0;
}

extern "C" int hrp_conv_pointwise(const hrp_conv_desc* d) {
  hrp::PwPlan pw;
  return d ? hrp::pw_plan(*d, pw) : 0;
}

extern "C" int hrp_conv2d_fwd(const hrp_conv_desc* d, void* stream) {
  using namespace hrp;
  const int rc = conv_check(d);
  if (rc != HRP_OK) return rc;
  if (d->dtype == HRP_F32) return launch_conv_f32(*d, (hipStream_t)stream);
  if (d->dtype == HRP_F32X3) return launch_conv_f32x3(*d, (hipStream_t)stream);
  if (hrp_conv_rowstrip_channels(d)) return launch_conv_row(*d, (hipStream_t)stream);
  PwPlan pw;
  if (pw_plan(*d, pw)) return launch_conv_pw(*d, pw, (hipStream_t)stream);
  return launch_conv<bf16_t>(*d, (hipStream_t)stream);
}

#ifdef HRP_TIMELINE
extern "C" int hrp_debug_conv_timeline(void* dst, int nblocks, int clear) {
  if (dst) (void)hipMemcpyFromSymbol(dst, HIP_SYMBOL(hrp::g_conv_timeline), sizeof(unsigned long long) * 8 * nblocks);
  if (clear) {
    void* p = nullptr;
    (void)hipGetSymbolAddress(&p, HIP_SYMBOL(hrp::g_conv_timeline));
    (void)hipMemset(p, 0, sizeof(unsigned long long) * 8192 * 8);
  }
  return 0;
}
#endif
