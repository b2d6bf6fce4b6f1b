"""Stage-by-stage HIP vs oracle comparison on a golden render case (debug aid, GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from nerfmeshes_amd import hip_ops as ops
from oracle import nerf_oracle as O
from tests.helpers import golden_hparams, golden_weights, load_golden, specs_from_hparams

case = sys.argv[1] if len(sys.argv) > 1 else "render_fern_8x128"
g = load_golden(case); hp = golden_hparams(g); sc, sf, rs = specs_from_hparams(hp); wc, wf = golden_weights(g, hp)
near, far = map(float, g["bounds"])
o, d = torch.from_numpy(g["origins"]), torch.from_numpy(g["directions"])
c, f = O.render(wc, wf, sc, sf, rs, o, d, near, far)
desc = lambda s: dict(num_layers=s.num_layers, hidden_size=s.hidden_size, skip_step=s.skip_step, num_encoding_fn_xyz=s.num_encoding_fn_xyz, num_encoding_fn_dir=s.num_encoding_fn_dir)
mc, mf = ops.HipMLP(wc, desc(sc), "cuda"), ops.HipMLP(wf, desc(sf), "cuda")
R = d.shape[0]
tc = ops.coarse_intervals(torch.linspace(0, 1, rs.num_coarse).cuda(), torch.tensor([near]).cuda(), torch.tensor([far]).cuda(), R, rs.lindisp)
print("t coarse equal", torch.equal(tc.cpu(), c["t"].contiguous()))
radc = mc.eval_rays(o.cuda(), d.cuda(), tc)
e = (radc.cpu() - c["radiance"]).abs()
print("rad coarse max err rgb %.3e sigma %.3e (sigma scale %.1f)" % (e[..., :3].max(), e[..., 3].max(), c["radiance"][..., 3].abs().max()))
compc = ops.composite(radc, tc, d.cuda(), white_background=rs.white_background)
print("coarse weights max err %.3e" % (compc["weights"].cpu() - c["weights"]).abs().max())
# sample pdf: ours on ours, ours on oracle's weights
u = torch.linspace(0, 1, rs.num_fine).cuda()
tf_own = ops.sample_pdf(tc, compc["weights"], u).cpu()
tf_orc = ops.sample_pdf(tc, c["weights"].cuda().contiguous(), u).cpu()
ref = f["t"]
for name, tf in (("hip(w_hip)", tf_own), ("hip(w_oracle)", tf_orc)):
    err = (tf - ref).abs()
    print(name, "t fine max err %.3e, rows>1e-5: %d, entries>1e-5: %d" % (err.max(), (err.max(-1).values > 1e-5).sum(), (err > 1e-5).sum()))
r = int((tf_own - ref).abs().max(-1).values.argmax())
print("worst ray", r, "positions", torch.nonzero((tf_own[r] - ref[r]).abs() > 1e-5).flatten().tolist())
print(" ours", tf_own[r][(tf_own[r] - ref[r]).abs() > 1e-5][:8].tolist())
print(" ref ", ref[r][(tf_own[r] - ref[r]).abs() > 1e-5][:8].tolist())
radf = mf.eval_rays(o.cuda(), d.cuda(), ref.cuda().contiguous())
e = (radf.cpu() - f["radiance"]).abs()
print("rad fine (on oracle t) max err rgb %.3e sigma %.3e" % (e[..., :3].max(), e[..., 3].max()))
compf = ops.composite(f["radiance"].cuda().contiguous(), ref.cuda().contiguous(), d.cuda(), white_background=rs.white_background)
for k in ("rgb_map", "acc_map", "disp_map", "weights"):
    print(" composite(oracle inputs)", k, "max err %.3e" % (compf[k].cpu() - f[k]).abs().max())
cb, fb = ops.render_rays(mc, mf, o.cuda(), d.cuda(), torch.tensor([near]), torch.tensor([far]), torch.linspace(0, 1, rs.num_coarse), torch.linspace(0, 1, rs.num_fine), lindisp=rs.lindisp, white_background=rs.white_background)
for k in ("rgb_map", "acc_map", "disp_map", "depth_map"):
    e = (fb[k].cpu() - f[k]).abs()
    print(" end-to-end", k, "max err %.3e at %s" % (e.max(), np.unravel_index(int(e.argmax()), e.shape)))
i = int((fb["disp_map"].cpu() - f["disp_map"]).abs().argmax())
print("ray", i, "acc", float(f["acc_map"][i]), float(fb["acc_map"][i]), "disp", float(f["disp_map"][i]), float(fb["disp_map"][i]))
