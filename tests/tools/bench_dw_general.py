"""The general weight-gradient kernel (nerf_dw_g.hip) shape by shape: what the planner picks, its time, and -- with --sweep --
every feasible geometry forced through NM_DW_FORCE, so that the planner's cost model can be checked against the machine.

    python tests/tools/bench_dw_general.py [--sweep] [--n 393216] [OUTxIN[:LDAxLDB] ...]
"""
import ctypes as C, itertools, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from nerfmeshes_amd import _lib

args = [a for a in sys.argv[1:] if not a.startswith("--")]
sweep = "--sweep" in sys.argv
n = int(sys.argv[sys.argv.index("--n") + 1]) if "--n" in sys.argv else 2048 * 192
if "--n" in sys.argv:
    args.remove(str(n))
shapes = []
for a in args or ["64x64", "64x39:64x40", "32x64", "100x100", "50x100", "128x128", "160x160", "256x256", "320x320", "320x63:320x64",
                  "160x320", "400x400", "512x512", "256x512"]:
    oi, _, ld = a.partition(":")
    o, i = (int(v) for v in oi.split("x"))
    lda, ldb = (int(v) for v in ld.split("x")) if ld else (o, i)
    shapes.append((o, lda, i, ldb))
os.environ.setdefault("NM_DW_GENERAL", "1")
lib = _lib.load()
dev = torch.device("cuda:0")
cus = torch.cuda.get_device_properties(dev).multi_processor_count
ptr = lambda x: C.c_void_p(x.data_ptr())


def run(o, lda, i, ldb, d, a, ws, dw, db, reps=5):
    def call():
        rc = lib.nm_weight_grad_ex(cus, ptr(d), o, lda, ptr(a), i, ldb, n, ptr(ws), ptr(dw), i, 0, ptr(db), None)
        if rc:
            raise RuntimeError(lib.nm_last_error().decode())
    call(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for x, y in ev:
        x.record(); call(); y.record()
    torch.cuda.synchronize()
    return min(x.elapsed_time(y) for x, y in ev)


out = {}
for o, lda, i, ldb in shapes:
    d = torch.randn(n, lda, device=dev); a = torch.randn(n, ldb, device=dev)
    need = max(int(lib.nm_weight_grad_workspace_bytes_ex(o, lda, i, ldb, cus)), 1 << 28)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    dw, db = torch.empty(o, i, device=dev), torch.empty(o, device=dev)
    os.environ.pop("NM_DW_FORCE", None)
    plan = (C.c_int32 * 8)()
    lib.nm_weight_grad_plan(o, lda, i, ldb, 1, cus, plan)
    ms = run(o, lda, i, ldb, d, a, ws, dw, db)
    ref = d[:, :o].double().t() @ a[:, :i].double()
    err = float((dw.double() - ref).abs().max() / ref.abs().max())
    row = {"plan[nba,nbb,wa,wb,wk,ta,tb,rows]": list(plan), "ms": round(ms, 4), "tflops": round(2.0 * n * o * i / ms / 1e9, 1),
           "frac_of_157.3": round(2.0 * n * o * i / ms / 1e9 / 157.3, 3), "GBps": round(4.0 * n * (lda + ldb) / ms / 1e6, 0), "rel_err": err}
    if sweep:
        res = []
        for nba, nbb in itertools.product((1, 2, 3, 4), repeat=2):
            for wa, wb, wk in ((4, 2, 1), (2, 4, 1), (8, 1, 1), (1, 8, 1), (2, 2, 2), (4, 1, 2), (1, 4, 2), (2, 1, 4), (1, 2, 4), (1, 1, 8)):
                for rows in (16, 32, 64, 128):
                    os.environ["NM_DW_FORCE"] = f"{nba},{nbb},{wa},{wb},{wk},{rows}"
                    try:
                        t = run(o, lda, i, ldb, d, a, ws, dw, db, reps=3)
                    except RuntimeError:
                        continue
                    e = float((dw.double() - ref).abs().max() / ref.abs().max())
                    res.append((round(t, 4), [nba, nbb, wa, wb, wk, rows], e))
        res.sort(key=lambda r: r[0])
        row["sweep_best5"] = res[:5]
        row["sweep_worst_err"] = max(r[2] for r in res)
        row["planner_over_best"] = round(ms / res[0][0], 3)
    out[f"{o}x{i}" + (f":{lda}x{ldb}" if (lda, ldb) != (o, i) else "")] = row
    print(f"{o}x{i}", json.dumps(row), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"n": n, "cus": cus, "shapes": out}, open(os.path.join(ROOT, "gpurun_out", "dw_general.json"), "w"), indent=1)
