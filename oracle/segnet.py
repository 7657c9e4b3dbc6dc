"""Oracle of the segmentation-mask network of BASELINE config 5 (TEST INFRASTRUCTURE).

Restates what ``seg_mask_inference.forward`` computes (reference lib/models/ctrnet/mask_inference.py:44-57):

  preprocess   np.uint8(img) -> PIL ``Image.resize`` to half size (Pillow's default filter for RGB images: BICUBIC with the
               support scaled by the reduction factor, two 8-bit passes in 22-bit fixed point) -> ToTensor (/ 255) -> Normalize
               (ImageNet mean / std)                                                      (mask_inference.py:44-49)
  network      ``KeyPointSegNet`` = torchvision ``deeplabv3_resnet50`` (ResNet-50 trunk with layer3 / layer4 dilated: output
               stride 8; ASPP with rates 12 / 24 / 36 + image pooling; 3x3 conv; 1x1 conv to ONE class), bilinear up-sampling to
               the input size (keypoint_seg_resnet.py:103-149), sigmoid (CtRNet.py:102-111).  The key-point branch
               (read_out + soft-argmax) is computed and discarded by ``inference_batch_images_onlyseg``; it is not restated.

PARITY STATUS.  torchvision (reference pin 0.14.1) and the authors' checkpoint are absent from the reference tree and from the
image: the network half is restated from torchvision's published architecture (torchvision/models/segmentation/deeplabv3.py,
torchvision/models/resnet.py) with its state-dict key names and is **parity unpinned**.  The resize half is pinned: Pillow is
installed here (12.2; the reference pins 9.5, same algorithm), and tests/test_oracle_golden.py checks ``pil_resize_half`` bit for
bit against ``PIL.Image.resize`` itself.
"""
import numpy as np
import torch
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
PRECISION_BITS = 32 - 8 - 2          # Pillow src/libImaging/Resample.c


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size, out_size):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the bicubic filter over the whole axis.
    -> (xmin [out], count [out], coeffs [out, ksize] int32 in 2^-22 units)."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 2.0 * fscale
    ksize = int(np.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int32)
    cnt = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / fscale
        lo = int(center - support + 0.5)
        lo = max(lo, 0)
        hi = int(center + support + 0.5)
        hi = min(hi, in_size)
        n = hi - lo
        w = np.array([_bicubic((x + lo - center + 0.5) * ss) for x in range(n)], dtype=np.float64)
        tot = w.sum()
        if tot != 0.0:
            w = w / tot
        q = np.where(w < 0, (-0.5 + w * (1 << PRECISION_BITS)).astype(np.int64), (0.5 + w * (1 << PRECISION_BITS)).astype(np.int64))
        xmin[xx], cnt[xx] = lo, n
        kk[xx, :n] = q
    return xmin, cnt, kk


def _pass(img, xmin, cnt, kk, axis):
    """one 8-bit resampling pass along `axis` of an [H, W, C] uint8 array"""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.zeros((len(xmin),) + src.shape[1:], np.int64)
    for o in range(len(xmin)):
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for j in range(cnt[o]):
            acc += src[xmin[o] + j] * int(kk[o, j])
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis).astype(np.uint8)


def pil_resize_half(img_u8, scale=0.5):
    """PIL.Image.fromarray(img).resize((int(W * scale), int(H * scale))) for an [H, W, 3] uint8 array: horizontal pass, then
    vertical pass (Resample.c: ImagingResampleInner)."""
    H, W = img_u8.shape[:2]
    ow, oh = int(W * scale), int(H * scale)
    t = _pass(img_u8, *resample_coeffs(W, ow), axis=1)
    return _pass(t, *resample_coeffs(H, oh), axis=0)


def preprocess(img_tensor, scale=0.5):
    """mask_inference.py:44-49 -> float32 [B, 3, H * scale, W * scale]"""
    arr = np.uint8(img_tensor.detach().cpu().numpy()).transpose(0, 2, 3, 1)
    out = []
    for a in arr:
        r = pil_resize_half(a, scale).astype(np.float32) / 255.0
        r = (r - np.array(IMAGENET_MEAN, np.float32)) / np.array(IMAGENET_STD, np.float32)
        out.append(torch.from_numpy(r.transpose(2, 0, 1).copy()))
    return torch.stack(out).float()


# ---- the network (torchvision deeplabv3_resnet50, replace_stride_with_dilation = [False, True, True]) ----------------------
def _bn(sd, k, x):
    return F.batch_norm(x, sd[k + ".running_mean"], sd[k + ".running_var"], sd[k + ".weight"], sd[k + ".bias"], False, 0.1, 1e-5)


def _bottleneck(sd, k, x, stride, dilation):
    """torchvision resnet.Bottleneck (v1.5: the stride sits on the 3x3 convolution)"""
    out = F.relu(_bn(sd, k + ".bn1", F.conv2d(x, sd[k + ".conv1.weight"])))
    out = F.relu(_bn(sd, k + ".bn2", F.conv2d(out, sd[k + ".conv2.weight"], stride=stride, padding=dilation, dilation=dilation)))
    out = _bn(sd, k + ".bn3", F.conv2d(out, sd[k + ".conv3.weight"]))
    if (k + ".downsample.0.weight") in sd:
        x = _bn(sd, k + ".downsample.1", F.conv2d(x, sd[k + ".downsample.0.weight"], stride=stride))
    return F.relu(out + x)


RESNET50_LAYERS = (3, 4, 6, 3)
ASPP_RATES = (12, 24, 36)


def layer_plan():
    """(layer name, blocks, [(stride, dilation) per block]) of the dilated ResNet-50 (torchvision resnet._make_layer with
    dilate = True for layer3 / layer4: the stride becomes 1, the FIRST block keeps the previous dilation)."""
    plan, dilation = [], 1
    for name, blocks, stride, dilate in (("layer1", 3, 1, False), ("layer2", 4, 2, False), ("layer3", 6, 2, True), ("layer4", 3, 2, True)):
        prev = dilation
        if dilate:
            dilation *= stride
            stride = 1
        plan.append((name, [(stride, prev)] + [(1, dilation)] * (blocks - 1)))
    return plan


def deeplab_forward(sd, x, prefix=""):
    """KeyPointSegNet's segmentation branch: logits [B, 1, H, W] at the input size.  sd: state dict with the reference's keys
    (``backbone.0.*`` = the ResNet, ``classifer.0.*`` = torchvision's DeepLabHead; the reference's spelling)."""
    s = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)} if prefix else sd
    H, W = x.shape[-2:]
    b = "backbone.0."
    h = F.relu(_bn(s, b + "bn1", F.conv2d(x, s[b + "conv1.weight"], stride=2, padding=3)))
    h = F.max_pool2d(h, 3, 2, 1)
    for name, blocks in layer_plan():
        for i, (stride, dil) in enumerate(blocks):
            h = _bottleneck(s, f"{b}{name}.{i}", h, stride, dil)
    c = "classifer.0."
    a = c + "0."                                                     # ASPP
    res = [F.relu(_bn(s, a + "convs.0.1", F.conv2d(h, s[a + "convs.0.0.weight"])))]
    for i, r in enumerate(ASPP_RATES):
        res.append(F.relu(_bn(s, f"{a}convs.{i + 1}.1", F.conv2d(h, s[f"{a}convs.{i + 1}.0.weight"], padding=r, dilation=r))))
    p = F.adaptive_avg_pool2d(h, 1)
    p = F.relu(_bn(s, a + "convs.4.2", F.conv2d(p, s[a + "convs.4.1.weight"])))
    res.append(F.interpolate(p, size=h.shape[-2:], mode="bilinear", align_corners=False))
    y = F.relu(_bn(s, a + "project.1", F.conv2d(torch.cat(res, 1), s[a + "project.0.weight"])))      # (Dropout: eval)
    y = F.relu(_bn(s, c + "2", F.conv2d(y, s[c + "1.weight"], padding=1)))
    y = F.conv2d(y, s[c + "4.weight"], s[c + "4.bias"])
    return F.interpolate(y, size=(H, W), mode="bilinear", align_corners=False)


def seg_mask_forward(sd, img_tensor, prefix="net.keypoint_seg_predictor.module.", scale=0.5):
    """seg_mask_inference.forward: [B, 3, H, W] images with values 0 .. 255 -> foreground probability [B, 1, H / 2, W / 2]"""
    x = preprocess(img_tensor, scale)
    return torch.sigmoid(deeplab_forward(sd, x, prefix))
