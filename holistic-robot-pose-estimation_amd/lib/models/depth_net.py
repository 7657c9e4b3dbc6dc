"""DepthNet (RootNet) on the plan runtime - drop-in for reference lib/models/depth_net.py:11-168.

``RootNet('hrnet32').forward(x, k_value)`` -> depth [B, 1] in the reference's unit (gamma * k_value, mm).
State-dict keys: ``backbone.*`` + ``depth_layer.{weight,bias}`` as in the reference; with ``add_fc`` also
``depth_fc1..5`` / ``depth_bn1..4`` (the residual MLP of depth_net.py:44-55, 113-120), with ``use_offset``
``offset_layer`` (depth_net.py:63-70, 127-131)."""
import torch
import torch.nn as nn

from hrpe_amd.plan import Term
from hrpe_amd.runtime import PlannedModule
from .backbones.HRnet import BatchNorm2d, Conv2d, get_hrnet
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
        if pred_xy:
            raise NotImplementedError("the pred_xy head (deconv + 2-D soft-argmax, ResNet trunks only in the reference) is "
                                      "off in every shipped config and not built")
        self.pred_xy, self.add_fc, self.use_offset = pred_xy, add_fc, use_offset
        self.input_shape = input_shape
        self.output_shape = (input_shape[0] // 4, input_shape[1] // 4)
        self.outplanes = 256
        if add_fc:      # depth_net.py:44-55 (nn.Linear weights are [out, in]: 1x1 convs on the pooled feature here;
            c = self.inplanes   # BatchNorm1d has BatchNorm2d's parameters / buffers)
            for i, (o, k) in enumerate([(c // 2, c), (c // 4, c // 2), (c // 4, c // 4), (c // 2, c // 4), (c, c // 2)], 1):
                setattr(self, f"depth_fc{i}", _Linear1x1(k, o))
                if i < 5:
                    setattr(self, f"depth_bn{i}", BatchNorm2d(o))
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
            feat = pb.avgpool(self.backbone.emit(pb, t))
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
        return ["x", "k_value"], [("dense", depth, (N, 1))], {"x": t}

    def forward(self, x, k_value):
        return self._run(x, k_value.reshape(-1, 1))[0]

    def init_weights(self):
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
