"""DepthNet (RootNet) on the plan runtime - drop-in for reference lib/models/depth_net.py:11-168.

``RootNet('hrnet32').forward(x, k_value)`` -> depth [B, 1] in the reference's unit (gamma * k_value, mm).
State-dict keys: ``backbone.*`` + ``depth_layer.{weight,bias}`` as in the reference."""
import torch
import torch.nn as nn

from hrpe_amd.runtime import PlannedModule
from .backbones.HRnet import Conv2d, get_hrnet
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
        if pred_xy or use_offset or add_fc:
            raise NotImplementedError("pred_xy / use_offset / add_fc heads are off in every shipped config and not built")
        self.pred_xy, self.add_fc, self.use_offset = pred_xy, add_fc, use_offset
        self.input_shape = input_shape
        self.output_shape = (input_shape[0] // 4, input_shape[1] // 4)
        self.outplanes = 256
        self.depth_layer = Conv2d(self.inplanes, 1, 1, bias=True)

    def _build(self, pb, x, k_value):
        N, Cc, H, W = x.shape
        kv = pb.vector_input("k_value", N, 1, dense=True)
        if self.backbone_name in ["hrnet", "hrnet32"]:
            t = pb.image_input("x", N, Cc, H, W, u8=x.dtype == torch.uint8)
            _, feat = self.backbone.emit(pb, t)
        else:   # depth_net.py:93-95: global average pooling of the ResNet feature map
            t = pb.image_input_s2d("x", N, Cc, H, W, u8=x.dtype == torch.uint8)
            feat = pb.avgpool(self.backbone.emit(pb, t))
        gamma = self.depth_layer.emit(pb, feat)          # 1x1 conv on [N,2048,1,1] == linear (depth_net.py:121-123)
        depth = pb.row_scale(gamma, kv)                  # depth = gamma * k_value (depth_net.py:125)
        return ["x", "k_value"], [("dense", depth, (N, 1))], {"x": t}

    def forward(self, x, k_value):
        return self._run(x, k_value.reshape(-1, 1))[0]

    def init_weights(self):
        nn.init.normal_(self.depth_layer.weight, std=0.001)
        nn.init.constant_(self.depth_layer.bias, 0)
        print("Initialized depth layer of RootNet.")


def get_rootnet(backbone, pred_xy=False, use_offset=False, add_fc=False, input_shape=(256, 256), **kwargs):
    model = RootNet(backbone, pred_xy, use_offset, add_fc, input_shape=(256, 256), **kwargs)
    model.init_weights()
    return model
