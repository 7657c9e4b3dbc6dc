"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol include/hrp.h
declares, ctypes structs match the C layout, URDF -> chain descriptor, module trees / state-dict keys,
error behaviour without a GPU, and the data-parallel gradient reducer over gloo (world_size 2)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import PANDA_URDF, ROOT

import hrpe_amd  # noqa: F401
from hrpe_amd import _native as nv


def header_functions():
    src = open(os.path.join(ROOT, "include", "hrp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hrp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = nv.lib()
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"libhrp_hip.so does not export {n}"
    # and every prototype the Python side binds is declared in the header
    for n in nv.PROTOTYPES:
        assert n in names, f"{n} bound in _native.py but not declared in include/hrp.h"
    assert lib.hrp_version() >= 100


def test_library_is_built_from_these_sources():
    """The loaded libhrp_hip.so embeds the sha256 of the sources it was compiled from (csrc/Makefile SRC_HASH)."""
    assert nv.lib().hrp_source_hash().decode() == nv.source_hash()


def test_ctypes_struct_sizes_match_c():
    """Compile a tiny C program against include/hrp.h and compare sizeof() with the ctypes mirrors."""
    prog = r'''
#include <stdio.h>
#include "hrp.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(hrp_conv_desc), sizeof(hrp_wgrad_desc), sizeof(hrp_pack_entry),
         sizeof(hrp_ew_input), sizeof(hrp_ew_desc), sizeof(hrp_ew_bwd_desc), sizeof(hrp_bn_entry), sizeof(hrp_fk_chain),
         sizeof(hrp_opt_tensor), sizeof(hrp_opt_chunk), sizeof(hrp_batch_info), sizeof(hrp_pose_loss_desc), sizeof(hrp_wgrad_fold_desc),
         sizeof(hrp_block_desc), sizeof(hrp_block_info), sizeof(hrp_regressor_step_desc), sizeof(hrp_linear_wgrad_desc), sizeof(hrp_copy_desc));
  return 0;
}'''
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "s.c"), "w").write(prog)
        exe = os.path.join(td, "s")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(td, "s.c"), "-o", exe], check=True)
        out = subprocess.run([exe], check=True, stdout=subprocess.PIPE, text=True).stdout.split()
    sizes = [int(v) for v in out]
    mirrors = [nv.ConvDesc, nv.WgradDesc, nv.PackEntry, nv.EwInput, nv.EwDesc, nv.EwBwdDesc, nv.BnEntry, nv.FkChain,
               nv.OptTensor, nv.OptChunk, nv.BatchInfo, nv.PoseLossDesc, nv.WgradFoldDesc,
               nv.BlockDesc, nv.BlockInfo, nv.RegStepDesc, nv.LinWgradDesc, nv.CopyDesc]
    assert sizes == [C.sizeof(m) for m in mirrors]
    assert (nv.REG_MAX_P, nv.REG_MAX_PROBLEMS, nv.LIN_WGRAD_MAX, nv.COPY_MAX) == (16, 4, 8, 16)      # HRP_REG_MAX_P, HRP_REG_MAX_PROBLEMS, HRP_LIN_WGRAD_MAX


def test_regressor_entry_points_reject_bad_descriptors_without_a_gpu():
    """hrp_regressor_step / hrp_linear_wgrad_batch / hrp_dropout_masks validate on the host before they launch."""
    d = nv.RegStepDesc()
    arr = (nv.RegStepDesc * 1)(d)
    assert nv.lib().hrp_regressor_step(arr, 1, None) == -1                       # M == 0
    assert nv.lib().hrp_regressor_step(arr, 5, None) == -1 and b"problems" in nv.lib().hrp_last_error()
    d = nv.RegStepDesc()
    d.M, d.P, d.K, d.N = 4, 17, 8, 8
    assert nv.lib().hrp_regressor_step((nv.RegStepDesc * 1)(d), 1, None) == -1  # state wider than HRP_REG_MAX_P
    d.P, d.K = 0, 6
    d.w = d.out = d.a = 16
    d.w_sn, d.w_sk, d.out_pitch = 6, 1, 8
    assert nv.lib().hrp_regressor_step((nv.RegStepDesc * 1)(d), 1, None) == -1 and b"K % 4" in nv.lib().hrp_last_error()
    g = nv.LinWgradDesc()
    assert nv.lib().hrp_linear_wgrad_batch((nv.LinWgradDesc * 1)(g), 1, None) == -1
    assert nv.lib().hrp_dropout_masks(None, 16, 0.5, None, 0, None) == -1
    c = nv.CopyDesc()
    assert nv.lib().hrp_copy_cols_batch((nv.CopyDesc * 1)(c), 1, None) == -1                  # no destination
    assert nv.lib().hrp_copy_cols_batch((nv.CopyDesc * 1)(c), 17, None) == -1 and b"problems" in nv.lib().hrp_last_error()
    c.dst, c.rows, c.cols, c.dst_pitch, c.accumulate = 16, 2, 3, 3, 1
    assert nv.lib().hrp_copy_cols_batch((nv.CopyDesc * 1)(c), 1, None) == -1 and b"accumulates nothing" in nv.lib().hrp_last_error()


def test_bad_descriptor_is_rejected_without_a_gpu():
    d = nv.ConvDesc()
    assert nv.lib().hrp_conv2d_fwd(C.byref(d), None) == -1
    assert b"null" in nv.lib().hrp_last_error()
    with pytest.raises(nv.HrpError):
        nv.call("hrp_conv2d_fwd", C.byref(d), None)


def test_no_cpu_fallback():
    """The product path refuses CPU tensors instead of silently computing with torch."""
    from hrpe_amd.lib.models.backbones.HRnet import BasicBlock
    from hrpe_amd.lib.utils.transforms import point_projection_from_3d_tensor
    with pytest.raises(nv.HrpError):
        BasicBlock(32, 32)(torch.zeros(1, 32, 8, 8))
    with pytest.raises(nv.HrpError):
        point_projection_from_3d_tensor(torch.eye(3)[None], torch.ones(1, 2, 3))


def test_urdf_chain_descriptor_matches_oracle_tree():
    from hrpe_amd.lib.dataset.const import JOINT_NAMES, LINK_NAMES
    from hrpe_amd.lib.utils.urdf_robot import URDFRobot, parse_chain
    from oracle import fk
    ch, names = parse_chain(PANDA_URDF, LINK_NAMES["panda"])
    tree = fk.Tree(PANDA_URDF)
    assert names == JOINT_NAMES["panda"] == [j["name"] for j in tree.actuated]
    assert ch.dof == 8 and ch.nkp == 7 and ch.njoints == len(tree.joints)
    # parents precede children; the mimic finger follows finger_joint1's column
    for j in range(ch.njoints):
        assert ch.parent[j] < j
    cols = [ch.cfg[j] for j in range(ch.njoints)]
    assert sorted(c for c in cols if c >= 0) == [0, 1, 2, 3, 4, 5, 6, 7, 7]
    # origin of panda_joint1 (z = 0.333) and keypoint frames
    o = [ch.origin[0][i] for i in range(12)]
    assert abs(o[11] - 0.333) < 1e-7 and abs(o[0] - 1.0) < 1e-7
    assert ch.kp_frame[0] == -1 and all(ch.kp_frame[k] >= 0 for k in range(1, 7))
    robot = URDFRobot("panda", urdf_path=PANDA_URDF)
    assert robot.dof == 8 and robot.link_names == LINK_NAMES["panda"]
    with pytest.raises(nv.HrpError):
        robot.get_keypoints_only_fk(torch.zeros(1, 8))      # CPU tensor: no fallback


def test_urdf_chain_descriptors_kuka_baxter():
    """Serial 7-DoF chain and the Baxter tree: actuated-joint column order (reference urdf.py:3795-3813), key-point
    frames and offsets (urdf_robot.py:52-74) as the FK kernel will see them."""
    import os
    from hrpe_amd.lib.dataset.const import JOINT_NAMES, LINK_NAMES
    from hrpe_amd.lib.utils.urdf_robot import URDFRobot
    from oracle import fk
    assets = os.path.dirname(PANDA_URDF)
    kuka = URDFRobot("kuka", urdf_path=os.path.join(assets, "kuka_kinematics.urdf"))
    assert kuka.dof == 7 and kuka.nkp == 8 and kuka.link_names == LINK_NAMES["kuka"]
    assert float(kuka.offsets.abs().max()) == 0.0 and kuka.chain.kp_frame[0] == -1
    bax = URDFRobot("baxter", urdf_path=os.path.join(assets, "baxter_kinematics.urdf"))
    orb = fk.Robot(os.path.join(assets, "baxter_kinematics.urdf"), "baxter")
    assert bax.dof == 15 and bax.nkp == 17 and list(bax.actuated_joint_names) == JOINT_NAMES["baxter"]
    assert bax.link_names == orb.link_names and bax.link_names[:3] == ["base", "right_arm_mount", "left_arm_mount"]
    np.testing.assert_allclose(bax.offsets.reshape(17, 3).numpy(), orb.offsets.numpy(), atol=1e-7)
    ch = bax.chain
    assert ch.kp_frame[0] == -1 and all(ch.kp_frame[k] >= 0 for k in range(1, 17))
    for j in range(ch.njoints):
        assert ch.parent[j] < j
    # two arms hang off the torso: the tree has joints whose parent is not the previous joint
    assert sum(1 for j in range(1, ch.njoints) if ch.parent[j] != j - 1) > 2
    assert sorted(ch.cfg[j] for j in range(ch.njoints) if ch.cfg[j] >= 0) == list(range(15))
    with pytest.raises(NotImplementedError):
        URDFRobot("owi535")


def test_summary_add_pck_matches_reference():
    """lib/utils/metrics.py:116-162 (AUC by a 10 000-step Python loop) against the sort + searchsorted form, on the
    per-image errors the reference itself produced (tests/golden/golden_metrics.npz)."""
    import os
    from hrpe_amd.lib.utils.metrics import summary_add_pck
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_metrics.npz"))
    for tag in ("fk", "int"):
        alldis = {"dis3d": [], "dis2d": []}
        for i in range(3):
            alldis["dis3d"].extend(list(g[f"{tag}{i}:error3d"].astype(np.float32)))
            alldis["dis2d"].extend(list(g[f"{tag}{i}:error2d"].astype(np.float32)))
        s = summary_add_pck(alldis)
        keys = [k for k in g.files if k.startswith(f"summary_{tag}:")]
        assert len(keys) == len(s) == 22
        for k in keys:
            np.testing.assert_allclose(s[k.split(":", 1)[1]], float(g[k]), rtol=1e-6, atol=1e-9, err_msg=k)
        # tensors in, same numbers
        s2 = summary_add_pck({k: torch.tensor(np.array(v)) for k, v in alldis.items()})
        assert s2 == s


def test_module_surface_and_state_dict_contract():
    """Names / kwargs of the reference's lib.models surface and the state-dict sizes it documents
    (RootNet 1956 entries; full net = 2 HRNets + heads + 2 buffers)."""
    from hrpe_amd.lib.dataset.const import INITIAL_JOINT_ANGLE
    from hrpe_amd.lib.models.backbones.HRnet import PoseHighResolutionNet, get_hrnet, load_hrnet_cfg  # noqa: F401
    from hrpe_amd.lib.models.depth_net import RootNet, get_rootnet
    from hrpe_amd.lib.models.full_net import RootNetwithRegInt, get_rootNetwithRegInt_model
    rn = get_rootnet("hrnet32")
    sd = rn.state_dict()
    assert len(sd) == 1956 and "depth_layer.weight" in sd and "backbone.stage4.2.fuse_layers.3.0.2.0.weight" in sd
    assert sd["backbone.conv1.weight"].shape == (64, 3, 3, 3) and sd["depth_layer.weight"].shape == (1, 2048, 1, 1)
    assert isinstance(rn, RootNet)
    with pytest.raises(NotImplementedError):
        get_rootnet("vgg")

    class A(dict):
        __getattr__ = dict.__getitem__
    args = A(backbone_name="hrnet32", rootnet_backbone_name="hrnet32", other_image_size=256.0, use_rpmg=False, n_iter=4,
             p_dropout=0.5, reg_joint_map=False, joint_conv_dim=[], rotation_dim=6, direct_reg_rot=False,
             rot_iterative_matmul=False, fix_root=True, bbox_3d_shape=[1300, 1300, 1300], reference_keypoint_id=3,
             add_fc=False, multi_kp=False, kps_need_depth=None, pretrained_rootnet=None)
    init = {"robot_type": "panda", "pose_params": INITIAL_JOINT_ANGLE, "cam_params": np.eye(4), "init_pose_from_mean": True}
    m = get_rootNetwithRegInt_model(init, args)
    assert isinstance(m, RootNetwithRegInt)
    keys = set(m.state_dict())
    for k in ("reg_backbone.final_layer.weight", "rootnet_backbone.final_feat_layer.1.running_var", "fc_pose_1.weight",
              "fc_rot_2.bias", "decpose.weight", "decrot.bias", "depth_layer.bias", "init_pose", "init_rot"):
        assert k in keys
    assert m.state_dict()["fc_pose_1.weight"].shape == (1024, 2056) and m.state_dict()["fc_rot_1.weight"].shape == (1024, 2054)
    assert torch.allclose(m.init_rot, torch.tensor([[1.0, 0, 0, 0, 1, 0]]))
    assert torch.allclose(m.init_pose[0, 3], torch.tensor(-1.52715))
    with pytest.raises(ValueError):
        RootNetwithRegInt({**init, "robot_type": "ur5"}, args)
    with pytest.raises(NotImplementedError):
        get_rootNetwithRegInt_model(init, A(args, backbone_name="vgg"))


def test_k_values_and_loss_harness_on_cpu():
    """Caller-side pieces are plain tensor expressions: check them against the oracle's restatement."""
    from hrpe_amd.lib.core.function import compute_k_values
    fx = torch.tensor([400.0, 512.0]); fy = torch.tensor([380.0, 512.0])
    bb = torch.tensor([[10.0, 20, 110, 90], [0.0, 0, 50, 200]])
    k = compute_k_values(fx, fy, bb)
    ref = torch.sqrt(fx * fy * 1e6 / torch.tensor([100.0, 200.0]) ** 2)
    assert torch.allclose(k, ref)


def _ddp_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from hrpe_amd.parallel import GradAllReducer, broadcast_module, init_distributed
    r, w, _ = init_distributed(backend="gloo")
    lin = torch.nn.Linear(4, 3)
    torch.manual_seed(100 + rank)
    with torch.no_grad():
        lin.weight.normal_()
    broadcast_module(lin)                     # everyone takes rank 0's parameters
    flat = torch.arange(10_000, dtype=torch.float32) * (rank + 1)
    GradAllReducer(bucket_mb=0.01)([flat])    # several buckets
    # the overlapped form: some ranges first, their complement later, one finish (bench.py at N > 1)
    red = GradAllReducer(bucket_mb=0.01)
    flat2 = torch.arange(10_000, dtype=torch.float32) * (rank + 1)
    first = [(0, 3000), (5000, 2500)]
    rest = GradAllReducer.complement(first, flat2.numel())
    assert rest == [(3000, 2000), (7500, 2500)]
    works = red.start(flat2, first)
    works += red.start(flat2, rest)
    red.finish(works, [flat2])
    assert torch.equal(flat, flat2)
    # k groups of ranges started one after the other (the k-cut backward of bench.py), tail last, one finish
    flat3 = torch.arange(10_000, dtype=torch.float32) * (rank + 1)
    groups = [[(0, 1000)], [(1000, 2500), (6000, 1000)], [(3500, 2500)]]
    tail = GradAllReducer.complement([r for g in groups for r in g], flat3.numel())
    assert tail == [(7000, 3000)]
    works = []
    for g in groups:
        works += red.start(flat3, g)
    works += red.start(flat3, tail)
    red.finish(works, [flat3])
    assert torch.equal(flat, flat3)
    # bf16 payload: the sum is formed in bf16, the mean in fp32 in the arena
    flat4 = torch.arange(10_000, dtype=torch.float32) * (rank + 1)
    GradAllReducer(bucket_mb=0.01, payload="bf16")([flat4])
    assert (flat4 - flat).abs().max().item() <= 2.0 ** -6 * flat.abs().max().item() and not torch.equal(flat4, flat)      # (three bf16 roundings)
    q.put((rank, lin.weight.detach().numpy().copy(), flat[:5].numpy().copy(), flat[-1].item()))
    dist.destroy_process_group()


def test_data_parallel_reducer_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + os.getpid() % 2000
    ps = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in ps], key=lambda t: t[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][1], res[1][1])                    # broadcast
    mean = np.arange(5, dtype=np.float32) * 1.5                    # (1x + 2x) / 2
    assert np.allclose(res[0][2], mean) and np.allclose(res[1][2], mean)
    assert abs(res[0][3] - 9999 * 1.5) < 1e-3


def test_fdiv16_arithmetic_is_exact():
    """csrc/hrp_common.h fdiv16: (int)((float(v) + 0.5f) * (1.0f / d)) == v // d for 0 <= v < 65536 - the
    plan arithmetic of the conv kernels relies on it (same fp32 operations restated in numpy)."""
    v = np.arange(65536, dtype=np.float32)
    ds = list(range(1, 700)) + [1023, 1024, 1025, 1296, 2047, 4095, 4096, 10000, 40000, 65535]
    for d in ds:
        inv = np.float32(1.0) / np.float32(d)
        q = ((v + np.float32(0.5)) * inv).astype(np.int32)
        assert np.array_equal(q, (np.arange(65536) // d).astype(np.int32)), d


def test_plan_lanes_bookkeeping():
    """plan.py lanes: flat parallel blocks fork / join around their ops, the backward mirrors them, and the
    concurrency predicate only fires for different lanes of ONE block (host logic, no GPU needed)."""
    import torch
    from hrpe_amd import plan as P
    assert P.lanes_concurrent(((1, 0),), ((1, 1),))
    assert not P.lanes_concurrent(((1, 0),), ((1, 0),))
    assert not P.lanes_concurrent(((1, 0),), ((2, 1),))          # different (sequential) blocks
    assert not P.lanes_concurrent((), ((1, 1),))                   # outside any block
    assert P.lanes_concurrent(((1, 0), (3, 0)), ((1, 1), (4, 2)))  # split in the outer block

    pl = P.Plan(torch.device("cpu"), torch.float32, True, True)
    pb = P.PlanBuilder(pl)
    ran = []
    pl.fwd.append(lambda s: ran.append("pre"))
    with pb.parallel(3) as par:
        for i in range(3):
            with par.lane(i):
                pl.fwd.append(lambda s, i=i: ran.append(f"lane{i}"))
                pb.bwd_stack.append(lambda i=i: pl.bwd.append(lambda s, i=i: ran.append(f"b{i}")))
                assert pl.lane_path == ((1, i),)
    pl.fwd.append(lambda s: ran.append("post"))
    kinds = [(e.lane, getattr(e.op, "kind", None)) for e in pl.fwd]
    assert kinds[0] == (0, None) and kinds[1] == (None, "fork") and kinds[-2] == (None, "join") and kinds[-1] == (0, None)
    assert [e.lane for e in pl.fwd[2:5]] == [0, 1, 2] and pl.n_lanes == 3
    assert [e.path for e in pl.fwd[2:5]] == [((1, 0),), ((1, 1),), ((1, 2),)]
    # a tensor whose gradient is written from two concurrent lanes is rejected at build time
    t = P.TensorH(pl, 1, 1, 1, 8, torch.float32, buf=torch.zeros(8))
    with pb.parallel(2) as par:
        with par.lane(0):
            t.take_grad_slot()
        with par.lane(1):
            with pytest.raises(RuntimeError):
                t.take_grad_slot()
    # backward: reverse order, the forward join becomes a fork and vice versa
    for lane, path, emit in reversed(pb.bwd_stack):
        if lane is None:
            list.append(pl.bwd, P.Entry(None, (), emit))
        else:
            pl.cur_lane, pl.lane_path = lane, path
            emit()
    bk = [(e.lane, getattr(e.op, "kind", None)) for e in pl.bwd]
    first = next(i for i, (l, k) in enumerate(bk) if k == "fork" and bk[i + 1][0] == 2)
    assert [l for l, _ in bk[first + 1:first + 4]] == [2, 1, 0] and bk[first + 4] == (None, "join")


def _fake_conv(y, ntaps=9, cin=32, cout=32, hw=64):
    d = nv.ConvDesc()
    d.x, d.w, d.y = 0x10000, 0x20000, y
    d.dtype, d.N, d.H, d.W, d.Cin, d.x_pitch = nv.HRP_BF16, 2, hw, hw, cin, cin
    d.Ho, d.Wo, d.Cout, d.y_H, d.y_W, d.y_pitch, d.res_pitch = hw, hw, cout, hw, hw, cout, cout
    d.out_stride, d.in_stride, d.ntaps, d.w_ntaps, d.w_cout_pad = 1, 1, ntaps, ntaps, (cout + 31) // 32 * 32
    k = 0
    for a in range(-1, 2):
        for b in range(-1, 2):
            if k < ntaps:
                d.dy[k], d.dx[k], d.wtap[k] = (a, b, k) if ntaps == 9 else (0, 0, k)
                k += 1
    return d


def test_plan_lockstep_merge_into_batched_launches():
    """plan.py merged mode (host logic): the lanes of a parallel block are walked in lock step, launches of one
    family / tap count at the same position fold into one batched launch, every lane keeps its own order, nested
    (virtual) blocks merge upwards, and two launches writing the same address never share a batch."""
    import torch
    from hrpe_amd import plan as P
    pl = P.Plan(torch.device("cpu"), torch.bfloat16, True, True)
    pb = P.PlanBuilder(pl)
    log = []
    pl.fwd.append(lambda s: log.append("pre"))
    with pb.parallel(3) as par:
        for i in range(3):
            with par.lane(i):
                pl.fwd.append(P.Launch("conv", _fake_conv(0x100000 * (i + 1))))                    # position 0: 3x3
                pl.fwd.append(lambda s, i=i: log.append(f"misc{i}"))                               # position 1: not batchable
                pl.fwd.append(P.Launch("conv", _fake_conv(0x900000 + 0x100000 * i, ntaps=1 if i else 9)))   # position 2: mixed taps
                if i == 2:   # a nested virtual block: two independent launches of this lane
                    with pb.parallel(2, virtual=True) as vp:
                        for j in range(2):
                            with vp.lane(j):
                                pl.fwd.append(P.Launch("conv", _fake_conv(0x2000000 + 0x100000 * j, ntaps=1)))
    pl.fwd.append(P.Launch("conv", _fake_conv(0x5000000)))
    mode = P.PLAN_MODE
    P.PLAN_MODE = "merged"          # every block virtual: one stream
    flat = [e.op for e in pl._flatten(pl.fwd)]
    kinds = [type(op).__name__ if not callable(op) or isinstance(op, (P.Launch, P.BatchLaunch)) else "fn" for op in flat]
    # (the lanes' sequences differ - lane 2 has the nested block -, so they merge by their heads: the largest group first)
    assert kinds == ["fn", "BatchLaunch", "fn", "fn", "fn", "BatchLaunch", "Launch", "BatchLaunch", "Launch"], kinds
    assert [len(op.items) for op in flat if isinstance(op, P.BatchLaunch)] == [3, 2, 2]
    assert flat[6].desc.ntaps == 9 and {it.desc.ntaps for it in flat[5].items} == {1}
    # lanes whose sequences are offset against each other find their partners: [3x3, 1x1] next to [1x1] is two launches
    # (by position it would be three)
    pl3 = P.Plan(torch.device("cpu"), torch.bfloat16, True, True)
    pb3 = P.PlanBuilder(pl3)
    with pb3.parallel(2, virtual=True) as par:
        with par.lane(0):
            pl3.fwd.append(P.Launch("conv", _fake_conv(0x100000)))
            pl3.fwd.append(P.Launch("conv", _fake_conv(0x200000, ntaps=1)))
        with par.lane(1):
            pl3.fwd.append(P.Launch("conv", _fake_conv(0x300000, ntaps=1)))
    ops3 = [e.op for e in pl3._flatten(pl3.fwd)]
    assert [type(op).__name__ for op in ops3] == ["Launch", "BatchLaunch"] and ops3[0].desc.ntaps == 9 and len(ops3[1].items) == 2
    # the same output twice at one position: two launches, not one batch
    pl2 = P.Plan(torch.device("cpu"), torch.bfloat16, True, True)
    pb2 = P.PlanBuilder(pl2)
    with pb2.parallel(2) as par:
        for i in range(2):
            with par.lane(i):
                pl2.fwd.append(P.Launch("conv", _fake_conv(0x100000)))
    assert [type(e.op).__name__ for e in pl2._flatten(pl2.fwd)] == ["Launch", "Launch"]
    # batching off: the same order, one by one
    P.BATCHING = False
    try:
        ops = [e.op for e in pl._flatten(pl.fwd)]
        assert all(not isinstance(op, P.BatchLaunch) for op in ops) and len(ops) == 13
    finally:
        P.BATCHING = True
    # hybrid: the block that asked for streams keeps its three lanes (fork / join markers around them), only the
    # virtual block inside lane 2 is merged
    P.PLAN_MODE = "hybrid"
    try:
        ents = pl._flatten(pl.fwd)
    finally:
        P.PLAN_MODE = mode
    marks = [getattr(e.op, "kind", None) for e in ents if e.lane is None]
    assert marks == ["fork", "join"]
    assert [e.lane for e in ents if e.lane is not None] == [0] + [0] * 3 + [1] * 3 + [2] * 4 + [0]
    assert sum(isinstance(e.op, P.BatchLaunch) for e in ents) == 1


def _fake_wgrad(dw, ntaps=9, cin=32, cout=32, hw=32, N=4):
    g = nv.WgradDesc()
    g.x, g.dy, g.dw, g.dtype = 0x10000, 0x20000, dw, nv.HRP_BF16
    g.N, g.H, g.W, g.Cin, g.x_pitch = N, hw, hw, cin, cin
    g.Ho, g.Wo, g.Cout, g.dy_pitch = hw, hw, cout, cout
    g.in_stride, g.ntaps, g.dw_cin, g.accumulate = 1, ntaps, cin, 1
    k = 0
    for a in range(-1, 2):
        for b in range(-1, 2):
            if k < ntaps:
                g.dy_t[k], g.dx_t[k] = (a, b) if ntaps == 9 else (0, 0)
                k += 1
    return g


def test_plan_regroups_weight_gradients_and_defers_their_folds():
    """plan.py host logic of round 2: the weight-gradient launches of a lane are regrouped into batches of one tap count
    wherever they sit (_sink_wgrads), run in phase 1 and are folded by HRP_BATCH_WGRAD_FOLD launches before the lane
    joins / at the end (_insert_folds); a launch marked `reserved` (its gradient is read right away) stays put."""
    import torch
    from hrpe_amd import plan as P
    pl = P.Plan(torch.device("cpu"), torch.bfloat16, True, True)
    P.PlanBuilder(pl)
    log = []
    ents = []
    for i in range(5):
        ents.append(P.Entry(0, (), P.Launch("wgrad", _fake_wgrad(0x100000 * (i + 1), ntaps=9 if i != 2 else 1))))
        ents.append(P.Entry(0, (), lambda s, i=i: log.append(i)))
    stay = _fake_wgrad(0x900000)
    stay.reserved = 1
    ents.insert(4, P.Entry(0, (), P.Launch("wgrad", stay)))
    saved = P.WGRAD_SINK
    P.WGRAD_SINK = 3
    try:
        out = pl._sink_wgrads(ents)
    finally:
        P.WGRAD_SINK = saved
    kinds = ["B%d" % len(e.op.items) if isinstance(e.op, P.BatchLaunch) else ("W" if isinstance(e.op, P.Launch) else "f") for e in out]
    # fn, fn, [reserved wgrad in place], fn, the batch of three 3x3 problems where the third arrived, fn, fn, then the rest
    assert kinds == ["f", "f", "W", "f", "B3", "f", "f", "W", "W"], kinds
    assert out[2].op.desc.reserved == 1 and {it.desc.ntaps for it in out[4].op.items} == {9}
    assert sorted(e.op.desc.ntaps for e in out[-2:]) == [1, 9]
    # phase 1 + deferred folds: a fold launch of the four deferred problems at the end, none for the reserved one
    for e in out:
        if isinstance(e.op, (P.Launch, P.BatchLaunch)):
            for it in e.op.launches():
                it.desc.phase = 0 if it.desc.reserved else 1
                need = int(nv.lib().hrp_wgrad_workspace_bytes(C.byref(it.desc)))
                it.desc.workspace, it.desc.workspace_bytes = 0x40000000, 1 << 30
                assert need > 0
    for e in out:
        if isinstance(e.op, P.BatchLaunch):
            e.op.prepare()
            assert e.op.info.grid2 == 0 and len(e.op.fold_descs()) == 3
    folded = pl._insert_folds(out)
    fam = [e.op.fam if isinstance(e.op, (P.Launch, P.BatchLaunch)) else "f" for e in folded]
    assert fam[:-1] == ["f" if k == "f" else "wgrad" for k in kinds] and fam[-1] == "wgrad_fold"
    fold = folded[-1].op
    assert len(fold.items) == 5 and fold.info.family == nv.BATCH_WGRAD_FOLD and fold.info.grid > 0
    assert {it.desc.dw for it in fold.items} == {0x100000 * (i + 1) for i in range(5)}


def test_wgrad_fold_descriptor_and_refusals_are_host_only():
    lib = nv.lib()
    g = _fake_wgrad(0x100000, cin=64, cout=64, hw=16, N=8)
    need = int(lib.hrp_wgrad_workspace_bytes(C.byref(g)))
    g.workspace, g.workspace_bytes, g.phase = 0x40000000, need, 1
    f = nv.WgradFoldDesc()
    assert lib.hrp_wgrad_fold_desc_of(C.byref(g), C.byref(f)) == 0
    assert f.G >= 1 and f.pairs == 4 and f.nte == 9 and f.nb == 1 and f.dw == 0x100000 and f.accumulate == 1
    assert f.G * f.pairs * f.nte * 1024 * 4 == need
    g.workspace_bytes = need - 4            # too small: the launch takes the atomics path, nothing to fold
    assert lib.hrp_wgrad_fold_desc_of(C.byref(g), C.byref(f)) == 0 and f.G == 0
    info = nv.BatchInfo()
    arr = (nv.WgradFoldDesc * 1)(f)
    host = (C.c_char * int(lib.hrp_batch_table_bytes(nv.BATCH_WGRAD_FOLD, 1)))()
    assert lib.hrp_batch_prepare(nv.BATCH_WGRAD_FOLD, arr, 1, host, C.byref(info)) == -1      # G == 0 is not foldable
    assert b"wgrad fold" in lib.hrp_last_error()
    # mixed phases in one batched weight-gradient launch are refused
    a, b = _fake_wgrad(0x100000), _fake_wgrad(0x200000)
    for d in (a, b):
        d.workspace, d.workspace_bytes = 0x40000000, 1 << 30
    a.phase = 1
    arr = (nv.WgradDesc * 2)(a, b)
    host = (C.c_char * int(lib.hrp_batch_table_bytes(nv.BATCH_WGRAD, 2)))()
    assert lib.hrp_batch_prepare(nv.BATCH_WGRAD, arr, 2, host, C.byref(info)) == -1 and b"mixed phases" in lib.hrp_last_error()


def test_plan_folds_bn_backward_reduce_into_the_data_gradient():
    """_fuse_bn_reduce (host logic): conv -> BN -> ReLU -> conv gives the data-gradient launch the BatchNorm operands
    and drops the reduce launch; an activation gradient with a second writer keeps its reduce pass."""
    import torch
    from hrpe_amd import plan as P
    pl = P.Plan(torch.device("cpu"), torch.bfloat16, True, True)
    P.PlanBuilder(pl)

    def act(out, mask, raw, nin=1):
        d = nv.EwDesc()
        d.nin, d.out, d.out_pitch, d.dtype = nin, out, 32, nv.HRP_BF16
        d.N, d.H, d.W, d.C, d.relu, d.mask, d.mask_pitch = 2, 64, 64, 32, 1, mask, 4
        d.inp[0].ptr, d.inp[0].pitch, d.inp[0].up, d.inp[0].mode = raw, 32, 1, nv.EW_BN_TRAIN
        return d

    def red(gout, mask, raw, sums):
        b = nv.EwBwdDesc()
        b.dout, b.dout_pitch, b.mask, b.mask_pitch, b.sums, b.dtype = gout, 32, mask, 4, sums, nv.HRP_BF16
        b.N, b.H, b.W, b.C, b.relu = 2, 64, 64, 32, 1
        b.inp.ptr, b.inp.pitch, b.inp.up, b.inp.mode = raw, 32, 1, nv.EW_BN_TRAIN
        return b

    # activation 1: gradient written by ONE conv -> fused.  activation 2: the conv accumulates onto an ew_app output.
    list.append(pl.fwd, P.Entry(0, (), P.Launch("ew_fwd", act(0x1000000, 0x1100000, 0x1200000))))
    list.append(pl.fwd, P.Entry(0, (), P.Launch("ew_fwd", act(0x2000000, 0x2100000, 0x2200000))))
    c1, c2 = _fake_conv(0x1300000), _fake_conv(0x2300000)
    r1, r2 = red(0x1300000, 0x1100000, 0x1200000, 0x1400000), red(0x2300000, 0x2100000, 0x2200000, 0x2400000)
    other = nv.EwBwdDesc()
    other.din2 = 0x2300000
    for op in (P.Launch("ew_app", other), P.Launch("conv", c1), P.Launch("ew_red", r1), P.Launch("ew_app", r1),
               P.Launch("conv", c2), P.Launch("ew_red", r2), P.Launch("ew_app", r2)):
        list.append(pl.bwd, P.Entry(0, (), op))
    # every gradient producer registers through TensorH.take_grad_slot: activation 1 has one producer (the conv), activation 2
    # two (the conv and the residual rider of an ew_app launch)
    import types
    pl.grad_owner[0x1300000] = types.SimpleNamespace(_grad_paths=[()])
    pl.grad_owner[0x2300000] = types.SimpleNamespace(_grad_paths=[(), ()])
    pl._fuse_bn_reduce()
    fams = [e.op.fam for e in pl.bwd]
    assert fams == ["ew_app", "conv", "ew_app", "conv", "ew_red", "ew_app"] and pl.counters["bn_reduce_fused"] == 1
    assert c1.bnb_x == 0x1200000 and c1.bnb_mask == 0x1100000 and c1.stats == 0x1400000 and c1.bnb_consts
    assert pl.fwd[0].op.desc.consts_out == c1.bnb_consts and not pl.fwd[1].op.desc.consts_out
    assert not c2.bnb_x and not c2.stats


def test_batch_prepare_is_host_only():
    """hrp_batch_prepare makes no HIP call: block ranges, per-problem tiles and the refusal paths can be checked here."""
    lib = nv.lib()
    descs = [_fake_conv(0x100000, cin=32, cout=32, hw=64), _fake_conv(0x200000, cin=256, cout=256, hw=8),
             _fake_conv(0x300000, cin=64, cout=64, hw=32)]
    arr = (nv.ConvDesc * 3)(*descs)
    info = nv.BatchInfo()
    nb = lib.hrp_batch_table_bytes(nv.BATCH_CONV, 3)
    assert nb > 0 and lib.hrp_batch_table_bytes(nv.BATCH_CONV, nv.BATCH_MAX + 1) == 0
    host = (C.c_char * nb)()
    assert lib.hrp_batch_prepare(nv.BATCH_CONV, arr, 3, host, C.byref(info)) == 0, lib.hrp_last_error()
    assert info.n == 3 and info.variant == 9 and info.blk0[0] == 0 and info.blk0[3] == info.grid
    # 2 images: 32 ch @64x64 -> row-strip kernel, 2 x 8 strips of 8 rows; 256 ch @8x8 -> whole-image kernel, one pair of
    # images x 2 blocks of 128 output channels; 64 ch @32x32 -> row-strip kernel, 2 x 4 strips
    assert sorted(info.blk0[i + 1] - info.blk0[i] for i in range(3)) == [2, 8, 16]
    assert [lib.hrp_conv_rowstrip_channels(C.byref(d)) for d in descs] == [32, 256, 64]
    assert lib.hrp_conv_rowstrip_channels(C.byref(_fake_conv(0x500000, cin=256, cout=128, hw=8))) == 0
    assert info.blk0[1] - info.blk0[0] == 2          # the deepest K loop goes first
    assert 0 < info.lds_bytes <= 160 * 1024
    # mixed tap counts are refused, and so is a half-filled last channel chunk
    bad = (nv.ConvDesc * 2)(descs[0], _fake_conv(0x400000, ntaps=1))
    assert lib.hrp_batch_prepare(nv.BATCH_CONV, bad, 2, host, C.byref(info)) != 0
    bad = (nv.ConvDesc * 1)(_fake_conv(0x400000, cin=8))
    assert lib.hrp_batch_prepare(nv.BATCH_CONV, bad, 1, host, C.byref(info)) != 0
    # weight gradients: the size query (no table) shares the launch's workgroups between the problems
    gs = []
    for cin, hw in ((32, 64), (128, 16)):
        g = nv.WgradDesc()
        g.x, g.dy, g.dw, g.dtype = 0x10000, 0x20000, 0x30000, nv.HRP_BF16
        g.N, g.H, g.W, g.Cin, g.x_pitch, g.Ho, g.Wo, g.Cout, g.dy_pitch = 64, hw, hw, cin, cin, hw, hw, cin, cin
        g.in_stride, g.ntaps, g.dw_cin, g.accumulate = 1, 9, cin, 1
        k = 0
        for a in range(-1, 2):
            for b in range(-1, 2):
                g.dy_t[k], g.dx_t[k] = a, b
                k += 1
        gs.append(g)
    arr = (nv.WgradDesc * 2)(*gs)
    assert lib.hrp_batch_prepare(nv.BATCH_WGRAD, arr, 2, None, C.byref(info)) == 0, lib.hrp_last_error()
    single = [lib.hrp_wgrad_workspace_bytes(C.byref(g)) for g in gs]
    # both are 3x3 stride-1 bf16 layers of the eight-wave program (its own launch of one 512-thread workgroup per CU): the
    # 32-channel problem writes one slab per eight waves, the 128-channel problem four per eight waves
    assert info.grid == 0 and 0 < info.grid3 <= 256 and info.lds_bytes3 <= 160 * 1024 and info.grid2 > 0
    assert 0 < info.ws_bytes[0] <= single[0] and 0 < info.ws_bytes[1] <= 2 * single[1]
    # a stride-2 layer with multiples of 64 channels runs the eight-wave program too (64-pixel tiles) ...
    gs[1].in_stride, gs[1].H, gs[1].W = 2, 32, 32
    arr = (nv.WgradDesc * 2)(*gs)
    assert lib.hrp_batch_prepare(nv.BATCH_WGRAD, arr, 2, None, C.byref(info)) == 0, lib.hrp_last_error()
    assert info.grid == 0 and 0 < info.grid3 <= 256
    # ... one that brings 32 input channels keeps the 32 x 32 program and its launch
    gs[1].Cin = gs[1].x_pitch = gs[1].dw_cin = 32
    arr = (nv.WgradDesc * 2)(*gs)
    assert lib.hrp_batch_prepare(nv.BATCH_WGRAD, arr, 2, None, C.byref(info)) == 0, lib.hrp_last_error()
    assert 0 < info.grid <= 512 and 0 < info.grid3 <= 256


def test_kernel_choice_and_workspace_queries_are_host_only():
    """Which kernel a convolution problem gets, and the workspace sizes of the ordered reductions, are host decisions that a
    caller can query without a GPU: hrp_conv_rowstrip_channels (3x3 C -> C layers with 4 KiB rows and their fused-BatchNorm
    fields), hrp_conv_pointwise (dense 1x1 layers above a pixel threshold), hrp_linear_workspace_bytes,
    hrp_colsum_workspace_bytes."""
    lib = nv.lib()
    # row-strip: the four branch shapes; fused fields are accepted there and nowhere else
    for c, hw in ((32, 64), (64, 32), (128, 16), (256, 8)):
        d = _fake_conv(0x400000, cin=c, cout=c, hw=hw)
        assert lib.hrp_conv_rowstrip_channels(C.byref(d)) == c
        d.res, d.res_mask = 0x600000, 0x700000                       # masked residual
        assert lib.hrp_conv_rowstrip_channels(C.byref(d)) == c
        d.relu = 1                                                     # ... not together with a ReLU epilogue
        assert lib.hrp_conv_rowstrip_channels(C.byref(d)) == 0
    d = _fake_conv(0x400000, cin=32, cout=32, hw=32)                  # 32 channels at 32 x 32: rows of 2 KiB -> tile program
    assert lib.hrp_conv_rowstrip_channels(C.byref(d)) == 0
    d = _fake_conv(0x400000, cin=64, cout=64, hw=32)
    d.pro_mode, d.pro_x2 = 2, 0x800000                                # the backward prologue needs its statistics
    assert lib.hrp_conv_rowstrip_channels(C.byref(d)) == 0
    d.pro_stats, d.pro_bsums, d.pro_gamma, d.pro_beta = 0x900000, 0x910000, 0x920000, 0x930000
    assert lib.hrp_conv_rowstrip_channels(C.byref(d)) == 64
    # pointwise: 64 -> 256 at 64 x 64 x 32 images = 131 072 pixels is in, half of that is not (default threshold)
    os.environ.pop("HRP_PW_MIN_PIXELS", None)
    d = _fake_conv(0x400000, ntaps=1, cin=64, cout=256, hw=64)
    d.N = 32
    assert lib.hrp_conv_pointwise(C.byref(d)) == 1
    d.N = 16
    assert lib.hrp_conv_pointwise(C.byref(d)) == 0
    d.N = 32
    d.bias = 0x600000
    assert lib.hrp_conv_pointwise(C.byref(d)) == 0
    d.bias = None
    d.Cin = d.x_pitch = 256                                            # 16 k-steps with an epilogue reduce: tile program
    d.Cout = d.y_pitch = d.res_pitch = d.w_cout_pad = 64
    assert lib.hrp_conv_pointwise(C.byref(d)) == 1
    d.bnb_x, d.bnb_x_pitch, d.bnb_mask, d.bnb_mask_pitch, d.bnb_consts, d.stats = 0x600000, 64, 0x700000, 8, 0x800000, 0x900000
    assert lib.hrp_conv_pointwise(C.byref(d)) == 0
    # ordered reductions: partial tiles of the reduction splits, padded to whole 64 x 32 tiles; forward and data gradient share
    assert lib.hrp_linear_workspace_bytes(64, 2056, 1024) == max(17 * 64 * 1024 * 4, 8 * 64 * 2080 * 4)
    assert lib.hrp_linear_workspace_bytes(5, 100, 6) == 64 * 128 * 4        # one split each way: max(1 x 64 x 32, 1 x 64 x 128) floats
    assert lib.hrp_linear_workspace_bytes(0, 8, 8) == 0
    assert lib.hrp_colsum_workspace_bytes(262144, 448) == 512 * 448 * 4       # at most 512 row groups
    assert lib.hrp_colsum_workspace_bytes(100, 32) == 2 * 32 * 4


def test_resnet_full_net_state_dict_keys():
    """backbone_name='resnet50' (the shipped full.yaml): key names / shapes of the reference's RootNetwithRegInt
    (reg_backbone.* 318 entries, deconv_layers.* 18, final_layer.*)."""
    from hrpe_amd.lib.dataset.const import INITIAL_JOINT_ANGLE
    from hrpe_amd.lib.models.full_net import RootNetwithRegInt

    class A(dict):
        __getattr__ = dict.__getitem__
    args = A(backbone_name="resnet50", rootnet_backbone_name="hrnet32", other_image_size=256.0, use_rpmg=False,
             n_iter=4, p_dropout=0.0, reg_joint_map=False, joint_conv_dim=[], rotation_dim=6, direct_reg_rot=False,
             rot_iterative_matmul=False, fix_root=True, bbox_3d_shape=[1300, 1300, 1300], reference_keypoint_id=3,
             add_fc=False, multi_kp=False, kps_need_depth=None, pretrained_rootnet=None)
    init = {"robot_type": "panda", "pose_params": INITIAL_JOINT_ANGLE, "cam_params": np.eye(4), "init_pose_from_mean": True}
    sd = RootNetwithRegInt(init, args).state_dict()
    assert sum(k.startswith("reg_backbone.") for k in sd) == 318
    assert sum(k.startswith("deconv_layers.") for k in sd) == 18
    assert len(sd) == 2308
    assert tuple(sd["reg_backbone.conv1.weight"].shape) == (64, 3, 7, 7)
    assert tuple(sd["reg_backbone.layer2.0.downsample.0.weight"].shape) == (512, 256, 1, 1)
    assert tuple(sd["deconv_layers.0.weight"].shape) == (2048, 256, 4, 4)
    assert tuple(sd["deconv_layers.7.running_var"].shape) == (256,)
    assert tuple(sd["final_layer.weight"].shape) == (448, 256, 1, 1)


def test_bench_self_launch_spawns_ranks_before_any_gpu_call(monkeypatch):
    """`python bench.py --gpus N` outside torch.distributed.run: the parent starts N ranks through
    `python -m torch.distributed.run` (127.0.0.1 rendezvous) without having initialised HIP itself, and a rank whose
    WORLD_SIZE disagrees with --gpus refuses to print a mislabelled line."""
    import importlib
    import subprocess as sp
    import torch
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    seen = {}

    class R:
        returncode = 0

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"], seen["cuda_init"] = cmd, env, torch.cuda.is_initialized()
        return R()
    monkeypatch.setattr(sp, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "127.0.0.1" in cmd
    assert cmd[cmd.index("--master-port") + 1].isdigit() and cmd[-6:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]
    assert seen["cuda_init"] is False and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # inside a launcher whose world size disagrees with --gpus: no line
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "mislabelled" in str(e.value)


def test_silhouette_oracle_known_answers_and_obj_loader(tmp_path):
    """oracle/silhouette.py (the torch restatement the HIP rasteriser is tested against; pytorch3d's algorithm, parity unpinned):
    a right triangle with legs of 20 px covers exactly the pixel centres strictly inside it (210 of them), a face behind the camera
    and a zero-area face render nothing, two overlapping faces give the same mask as one; and the .obj loader of
    hrpe_amd.lib.utils.mesh_renderer (host code) fans polygons and numbers vertices per file."""
    from oracle import silhouette as osil
    uv = torch.tensor([[[10.2, 10.2], [30.2, 10.2], [10.2, 30.2], [5.0, 5.0], [5.0, 5.0], [5.0, 5.0]]])
    z = torch.ones(1, 6)
    a = osil.soft_silhouette(uv, z, torch.tensor([[0, 1, 2]]), 40, 40)
    ys, xs = torch.meshgrid(torch.arange(40.0) + 0.5, torch.arange(40.0) + 0.5, indexing="ij")
    inside = (xs > 10.2) & (ys > 10.2) & ((xs - 10.2) + (ys - 10.2) < 20.0)
    assert int(inside.sum()) == 210
    assert torch.equal(a[0] > 0.5, inside) and float((a[0] * (1 - a[0])).abs().max()) < 1e-6       # binary at sigma = 1e-8
    assert torch.equal(osil.soft_silhouette(uv, z, torch.tensor([[0, 1, 2], [0, 2, 1]]), 40, 40) > 0.5, a > 0.5)
    assert float(osil.soft_silhouette(uv, z, torch.tensor([[3, 4, 5]]), 40, 40).max()) == 0.0       # zero area
    zb = z.clone()
    zb[0, 1] = -1.0
    assert float(osil.soft_silhouette(uv, zb, torch.tensor([[0, 1, 2]]), 40, 40).max()) == 0.0       # a vertex behind the camera
    from hrpe_amd.lib.utils.mesh_renderer import load_mesh_files
    p0, p1 = tmp_path / "a.obj", tmp_path / "b.obj"
    p0.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n")
    p1.write_text("v 0 0 1\nv 1 0 1\nv 0 1 1\nvn 0 0 1\nf 1//1 2//1 3//1\n")
    verts, links, faces = load_mesh_files([str(p0), str(p1)])
    assert verts.shape == (7, 3) and links.tolist() == [0, 0, 0, 0, 1, 1, 1]
    assert faces.tolist() == [[0, 1, 2], [0, 2, 3], [4, 5, 6]]


def test_plan_cache_key_tells_frozen_batchnorm_subsets_apart():
    """the plan-cache key of a training module with some BatchNorm modules in eval() was the COUNT of frozen
    modules - the regression trunk's and the DepthNet's BatchNorms (same count) shared a plan with the wrong bn.training baked in.
    The signature is the identity of the frozen set, and it sees a direct `m.training = False`."""
    import torch.nn as nn
    from hrpe_amd.runtime import PlannedModule
    from hrpe_amd.lib.models.backbones.HRnet import BatchNorm2d

    class Two(PlannedModule):
        def __init__(self):
            super().__init__()
            self.a = nn.ModuleList([BatchNorm2d(8) for _ in range(3)])
            self.b = nn.ModuleList([BatchNorm2d(8) for _ in range(3)])

    m = Two().train()
    none = m._frozen_bn_signature()
    assert none == 0
    for bn in m.a:
        bn.eval()
    sig_a = m._frozen_bn_signature()
    m.train()
    for bn in m.b:
        bn.eval()
    sig_b = m._frozen_bn_signature()
    assert sig_a != 0 and sig_b != 0 and sig_a != sig_b          # same count, different sets
    m.train()
    m.a[1].training = False                                        # no train() / eval() call: MODE_EPOCH does not move
    assert m._frozen_bn_signature() not in (0, sig_a, sig_b)
    m.eval()
    assert m._frozen_bn_signature() == 0                           # an eval-mode module: the inference plan, whatever the children say


def test_mask_network_refuses_a_missing_checkpoint():
    """the reference fails in torch.load when keypoint_seg_model_path does not exist (CtRNet.py:35); a silent random
    initialisation would let train_sim2real self-train against random masks.  Random weights are an explicit opt-in."""
    from hrpe_amd.lib.models.ctrnet.mask_inference import seg_mask_inference
    with pytest.raises(FileNotFoundError, match="keypoint_seg_model_path"):
        seg_mask_inference((600.0, 600.0, 320.0, 240.0), "azure")
    m = seg_mask_inference((600.0, 600.0, 320.0, 240.0), "azure", allow_random_init=True)
    assert any(k.startswith("net.keypoint_seg_predictor.module.") for k in m.state_dict())
