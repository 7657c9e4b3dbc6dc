import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import test_gpu_rowbw as T
R = T.R
nv = R.nvmod()
for shape in [(64, 5, 8), (64, 16, 32), (32, 16, 64)]:
    Cc, N, H = shape
    pb = T.build_block_problem(nv, Cc, N, H, 77, "g2")
    d, keep = T.device_operands(nv, pb)
    outs = []
    for it in range(8):
        keep["bs1"].zero_(); keep["y"].zero_()
        dw = torch.zeros(Cc * Cc * 9, device="cuda:0")
        q = T.rowbw_desc(nv, d, keep, pb, dw)
        T.run_rowbw(nv, [q])
        outs.append((keep["y"].clone(), dw.clone(), keep["bs1"].view(8, -1).sum(0).clone()))
    y0, w0, s0 = outs[0]
    for i, (y, w, s) in enumerate(outs[1:]):
        ds = (s - s0).abs()
        nz = torch.nonzero(ds > 1e-6).flatten().tolist()
        if nz: print("   nonzero diffs at", [(k, round(float(s[k] - s0[k]), 4)) for k in nz][:12])
        print(shape, i, "y equal", torch.equal(y, y0), "dw equal", torch.equal(w, w0), "stats maxdiff", float(ds.max()), "at", int(ds.argmax()), "of", float(s0.abs().max()))
