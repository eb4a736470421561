"""Developer probe: what would running the objective's two backward kernels (photometric, regularizer) SIDE BY SIDE buy?
Launches them on one stream (A B A B ...) and on two streams (A A A ... | B B B ...) with no cross-stream events in the loop,
at the headline image size, and prints both wall times per pair.  The difference is the ceiling of any fusion of the two."""
import ctypes as C
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussmart_amd import _lib
from gaussmart_amd.fused_regularizer import camera_kinv
from gaussmart_amd.synthetic import jittered_cameras

dev = torch.device("cuda:0")
H, W = 1080, 1920
L = _lib.lib()
cam = jittered_cameras(1, W, H, device=dev)[0]
kinv = camera_kinv(cam)
img, gt = torch.rand(3, H, W, device=dev), torch.rand(3, H, W, device=dev)
maps = torch.empty(3, 3, H, W, device=dev)
part = torch.empty(2, L.gsr_loss_num_partials(H, W), device=dev)
am = torch.rand(7, H, W, device=dev) + 0.1
dimg, dam = torch.empty_like(img), torch.empty_like(am)
scale = torch.ones(1, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def A(s): _lib.check(L.gsr_loss_backward(p(img), p(gt), p(maps), 3, H, W, 0.2, p(scale), p(dimg), C.c_void_p(s.cuda_stream)))
def B(s): _lib.check(L.gsr_regularizer_backward(p(am), H, W, kinv, 0.0, 0.05, 0.0, p(scale), p(dam), C.c_void_p(s.cuda_stream)))
_lib.check(L.gsr_loss_forward(p(img), p(gt), 3, H, W, p(maps), p(part[0]), C.c_void_p(s1.cuda_stream)))
torch.cuda.synchronize()
n = 300
def timed(fn):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e6
print("A alone            %.1f us" % timed(lambda: A(s1)))
print("B alone            %.1f us" % timed(lambda: B(s1)))
print("A B on one stream  %.1f us per pair" % timed(lambda: (A(s1), B(s1))))
print("A | B two streams  %.1f us per pair" % timed(lambda: (A(s1), B(s2))))
