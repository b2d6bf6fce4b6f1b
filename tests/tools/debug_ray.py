import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nerfmeshes_amd import hip_ops as ops, synthetic as S
from oracle import nerf_oracle as O
from tests.helpers import golden_hparams, golden_weights, load_golden, specs_from_hparams
g = load_golden("render_lego_scene"); hp = golden_hparams(g); sc, sf, rs = specs_from_hparams(hp); wc, wf = golden_weights(g, hp)
o, d = torch.from_numpy(g["origins"]), torch.from_numpy(g["directions"])
c, f = O.render(wc, wf, sc, sf, rs, o, d, 2.0, 6.0)
kw = dict(num_layers=8, hidden_size=256, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
m = ops.HipMLP(wc, kw, "cuda")
uc, uf = torch.linspace(0, 1, 64).cuda(), torch.linspace(0, 1, 128).cuda()
n = d.shape[0]
tc = ops.coarse_intervals(uc, torch.tensor([2.0]).cuda(), torch.tensor([6.0]).cuda(), n)
radc = m.eval_rays(o.cuda(), d.cuda(), tc); cc = ops.composite(radc, tc, d.cuda())
tf = ops.sample_pdf(tc, cc["weights"], uf)
radf = m.eval_rays(o.cuda(), d.cuda(), tf); ff = ops.composite(radf, tf, d.cuda())
err = (ff["rgb_map"].cpu() - f["rgb_map"]).abs().max(-1).values
r = int(err.argmax()); print("worst ray", r, "err", float(err[r]), "acc ref/ours", float(f["acc_map"][r]), float(ff["acc_map"][r]))
dt = (tf[r].cpu() - f["t"][r]).abs(); pos = torch.nonzero(dt > 1e-6).flatten().tolist()
print("t differs at", pos, [ (float(f["t"][r,p]), float(tf[r,p])) for p in pos[:6]])
for p in pos[:6] + [191]:
    print(" pos", p, "sigma ref %.4f ours %.4f  w ref %.4e ours %.4e" % (f["radiance"][r,p,3], radf[r,p,3], f["weights"][r,p], ff["weights"][r,p]))
wcw = c["weights"][r,1:-1]+1e-5; pdf = wcw/wcw.sum(); cdf = torch.cumsum(pdf,-1)
print("coarse pdf tail", pdf[-4:].tolist(), "cdf tail", cdf[-3:].tolist(), "coarse w max err", float((cc["weights"][r].cpu()-c["weights"][r]).abs().max()))
