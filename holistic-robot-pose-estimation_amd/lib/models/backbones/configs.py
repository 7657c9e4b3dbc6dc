"""Built-in HRNet topology tables (values of reference lib/models/backbones/configs/hrnet_w32.yaml:26-93
and hrnet_w48.yaml) so that ``get_hrnet`` does not depend on the caller's working directory."""


class AttrDict(dict):
    """dict with attribute access, recursively (stand-in for easydict.EasyDict)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    __setattr__ = __setitem__


def _stage(modules, channels):
    n = len(channels)
    return {"NUM_MODULES": modules, "NUM_BRANCHES": n, "BLOCK": "BASIC", "NUM_BLOCKS": [4] * n,
            "NUM_CHANNELS": list(channels), "FUSE_METHOD": "SUM"}


def _hrnet(width, pretrained):
    c = width
    return {
        "MODEL": {
            "INIT_WEIGHTS": True, "NAME": "pose_hrnet", "NUM_JOINTS": 7, "PRETRAINED": pretrained,
            "TARGET_TYPE": "gaussian", "IMAGE_SIZE": [256, 256], "HEATMAP_SIZE": [64, 64], "SIGMA": 2,
            "EXTRA": {
                "PRETRAINED_LAYERS": ["conv1", "bn1", "conv2", "bn2", "layer1", "transition1", "stage2",
                                      "transition2", "stage3", "transition3", "stage4", "incre_modules"],
                "FINAL_CONV_KERNEL": 1,
                "STAGE2": _stage(1, [c, 2 * c]),
                "STAGE3": _stage(4, [c, 2 * c, 4 * c]),
                "STAGE4": _stage(3, [c, 2 * c, 4 * c, 8 * c]),
            },
        }
    }


HRNET_CONFIGS = {
    "hrnet_w32": _hrnet(32, "./models/hrnet_w32-36af842e_roc.pth"),
    "hrnet_w48": _hrnet(48, "./models/hrnet_w48-8ef0771d.pth"),
}
