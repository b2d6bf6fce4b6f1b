"""Times nm_mlp_backward alone (8x256, 2048 rays x 192 samples); used for the PMC passes in profiles/r01_pmc_train_kernels.json."""
import os, sys, json, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nerfmeshes_amd import hip_ops, synthetic as S, train_ops as T, _lib
from nerfmeshes_amd._lib import MlpDeltas, MlpTape
kw = dict(num_layers=8, hidden_size=256, skip_step=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
mlp = hip_ops.HipMLP(S.make_scene_weights(**kw), kw, "cuda")
dev=torch.device("cuda:0"); R=2048
t = torch.sort(2.0 + 4.0 * torch.rand(R, 192, device=dev), dim=-1).values
o = torch.tensor([[0.,0.,4.]],device=dev); d=torch.nn.functional.normalize(torch.randn(R,3,device=dev),dim=-1)
n=R*192; rad,tape=T.forward_train(mlp,o,d,t); grad=torch.randn_like(rad)
lib=_lib.load(); f32=dict(dtype=torch.float32,device=dev)
dh,dfeat,dv,dlast=torch.empty(8,n,256,**f32),torch.empty(n,256,**f32),torch.empty(n,128,**f32),torch.empty(n,4,**f32)
ct = MlpTape(*[C.c_void_p(tape[k].data_ptr()) for k in ("h","feat","v","mask_h","mask_v")])
cd = MlpDeltas(*[C.c_void_p(x.data_ptr()) for x in (dh,dfeat,dv,dlast)])
st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(): lib.nm_mlp_backward(mlp.handle,n,C.byref(ct),C.c_void_p(rad.data_ptr()),C.c_void_p(grad.data_ptr()),C.byref(cd),st)
for _ in range(3): run()
torch.cuda.synchronize(); a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): run()
b.record(); torch.cuda.synchronize(); ms=a.elapsed_time(b)/10
print(json.dumps({"bwd_ms":ms,"tflops":2*(128*256+8*256*256)*n/ms/1e9}))
