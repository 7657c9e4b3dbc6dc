"""DepthNet (RootNet) on the plan runtime - drop-in for reference lib/models/depth_net.py:11-168.

``RootNet('hrnet32').forward(x, k_value)`` -> depth [B, 1] in the reference's unit (gamma * k_value, mm).
State-dict keys: ``backbone.*`` + ``depth_layer.{weight,bias}`` as in the reference; with ``add_fc`` also
``depth_fc1..5`` / ``depth_bn1..4`` (the residual MLP of depth_net.py:44-55, 113-120), with ``use_offset``
``offset_layer`` (depth_net.py:63-70, 127-131)."""
import torch
import torch.nn as nn

from hrpe_amd.plan import Term
from hrpe_amd.runtime import PlannedModule
from .backbones.HRnet import BatchNorm1d, BatchNorm2d, Conv2d, get_hrnet
from .backbones.Resnet import get_resnet


class RootNet(PlannedModule):
    def __init__(self, backbone, pred_xy=False, use_offset=False, add_fc=False, input_shape=(256, 256), **kwargs):
        super().__init__()
        self.backbone_name = backbone
        if backbone in ["hrnet", "hrnet32"]:
            self.backbone = get_hrnet(type_name=32, num_joints=7, depth_dim=1, pretrain=True,
                                      generate_feat=True, generate_hm=False)
            self.inplanes = 2048
        elif backbone in ["resnet34", "resnet50", "resnet"]:
            self.backbone = get_resnet(backbone)                       # depth_net.py:16-18
            self.inplanes = self.backbone.block.expansion * 512
        else:
            raise NotImplementedError
        if pred_xy and backbone in ["hrnet", "hrnet32"]:
            raise NotImplementedError("pred_xy needs the trunk's feature map: ResNet trunks only (the reference's forward "
                                      "references `fm`, which its HRNet branch never defines, depth_net.py:93-99)")
        self.pred_xy, self.add_fc, self.use_offset = pred_xy, add_fc, use_offset
        self.input_shape = input_shape
        self.output_shape = (input_shape[0] // 4, input_shape[1] // 4)
        self.outplanes = 256
        if pred_xy:      # depth_net.py:33-43, 72-90: three ConvTranspose2d(4, s2, p1) + BN + ReLU, then a 1x1 conv to one map
            from .full_net import ConvTranspose2d
            mods, cin = [], self.inplanes
            for _ in range(3):
                mods += [ConvTranspose2d(cin, self.outplanes), BatchNorm2d(self.outplanes), nn.Identity()]
                cin = self.outplanes
            self.deconv_layers = nn.Sequential(*mods)
            self.xy_layer = Conv2d(self.outplanes, 1, 1, bias=True)
        if add_fc:      # depth_net.py:44-55 (nn.Linear weights are [out, in]: 1x1 convs on the pooled feature here;
            c = self.inplanes   # BatchNorm1d has BatchNorm2d's parameters / buffers)
            for i, (o, k) in enumerate([(c // 2, c), (c // 4, c // 2), (c // 4, c // 4), (c // 2, c // 4), (c, c // 2)], 1):
                setattr(self, f"depth_fc{i}", _Linear1x1(k, o))
                if i < 5:
                    setattr(self, f"depth_bn{i}", BatchNorm1d(o))
        self.depth_layer = Conv2d(self.inplanes, 1, 1, bias=True)
        if use_offset:
            self.offset_layer = Conv2d(self.inplanes, 1, 1, bias=True)

    def _build(self, pb, x, k_value):
        N, Cc, H, W = x.shape
        kv = pb.vector_input("k_value", N, 1, dense=True)
        if self.backbone_name in ["hrnet", "hrnet32"]:
            t = pb.image_input("x", N, Cc, H, W, u8=x.dtype == torch.uint8)
            _, feat = self.backbone.emit(pb, t)
        else:   # depth_net.py:93-95: global average pooling of the ResNet feature map
            t = pb.image_input_s2d("x", N, Cc, H, W, u8=x.dtype == torch.uint8)
            fm = self.backbone.emit(pb, t)
            feat = pb.avgpool(fm)
        coords = None
        if self.pred_xy:         # depth_net.py:98-110: softmax over the 64 x 64 map, expected column and row index
            from hrpe_amd.plan import TensorH
            h = fm
            for i in (0, 3, 6):
                h = pb.act([Term(pb.deconv4x4s2(h, self.deconv_layers[i].weight, want_stats=pb.plan.training),
                                 self.deconv_layers[i + 1])], relu=True)
            xy = self.xy_layer.emit(pb, h)
            # one logit per pixel in a row of `pitch` (8) channels: the soft-argmax kernel reads whole 16-byte vectors of depth
            # bins, so the 7 padding channels are parked at -1e30 once (exp -> 0; the conv only ever writes channel 0)
            xy.buf.fill_(-1e30)
            uvd = pb.softargmax(xy, 1, xy.pitch, 0, False)       # u, v = expectation / size - 0.5
            uv = TensorH(pb.plan, N, 1, 1, 2, torch.float32, buf=uvd.buf, offset=uvd.offset, pitch=uvd.pitch, base=uvd)
            uv.requires_grad = uvd.requires_grad
            size = pb.constant(N, 2, 0.0)
            size.buf.view(N, 2)[:, 0], size.buf.view(N, 2)[:, 1] = float(xy.W), float(xy.H)
            half = pb.constant(N, 2, 0.0)
            half.buf.view(N, 2)[:, 0], half.buf.view(N, 2)[:, 1] = 0.5 * xy.W, 0.5 * xy.H
            coords = pb.act([Term(pb.row_scale(uv, size)), Term(half)], relu=False)      # coord = (u + 0.5) * size
        if self.add_fc:          # depth_net.py:113-120: feat += fc5(relu(bn4(fc4(.. relu(bn1(fc1(feat)))))))
            h = feat
            for i in range(1, 5):
                fc, bn = getattr(self, f"depth_fc{i}"), getattr(self, f"depth_bn{i}")
                h = pb.act([Term(pb.conv(h, fc.weight, fc.bias, want_stats=pb.plan.training), bn)], relu=True)
            feat = pb.conv(h, self.depth_fc5.weight, self.depth_fc5.bias, residual=feat)
        gamma = self.depth_layer.emit(pb, feat)          # 1x1 conv on [N,2048,1,1] == linear (depth_net.py:121-123)
        depth = pb.row_scale(gamma, kv)                  # depth = gamma * k_value (depth_net.py:125)
        if self.use_offset:      # depth_net.py:127-131: depth += 1000 * offset_layer(feat)   (offset in metres)
            depth = pb.row_scale(self.offset_layer.emit(pb, feat), pb.constant(N, 1, 1000.0), into=depth)
        if coords is not None:   # depth_net.py:133-135
            return ["x", "k_value"], [("dense", pb.cat_cols([coords, depth]), (N, 3))], {"x": t}
        return ["x", "k_value"], [("dense", depth, (N, 1))], {"x": t}

    def forward(self, x, k_value):
        return self._run(x, k_value.reshape(-1, 1))[0]

    def init_weights(self):
        if self.pred_xy:         # depth_net.py:140-151
            for m in self.deconv_layers:
                if hasattr(m, "weight") and m.weight.dim() == 4:
                    nn.init.normal_(m.weight, std=0.001)
            nn.init.normal_(self.xy_layer.weight, std=0.001)
            nn.init.constant_(self.xy_layer.bias, 0)
            print("Initialized deconv and xy layer of RootNet.")
        nn.init.normal_(self.depth_layer.weight, std=0.001)
        nn.init.constant_(self.depth_layer.bias, 0)
        print("Initialized depth layer of RootNet.")
        if self.use_offset:      # depth_net.py:157-162
            nn.init.normal_(self.offset_layer.weight, std=0.001)
            nn.init.constant_(self.offset_layer.bias, 0)
            print("Initialized offset layer of RootNet.")


class _Linear1x1(PlannedModule):
    """nn.Linear parameters ([out, in] weight + bias, torch's default initialisation) run as a 1x1 convolution of the
    [N, 1, 1, in] feature - the form whose epilogue gathers the BatchNorm statistics."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        lin = nn.Linear(in_features, out_features)
        self.weight, self.bias = nn.Parameter(lin.weight.detach().clone()), nn.Parameter(lin.bias.detach().clone())


def get_rootnet(backbone, pred_xy=False, use_offset=False, add_fc=False, input_shape=(256, 256), **kwargs):
    model = RootNet(backbone, pred_xy, use_offset, add_fc, input_shape=(256, 256), **kwargs)
    model.init_weights()
    return model
