import sys, torch
sys.path.insert(0, ".")
from msr3d_amd import _lib
from msr3d_amd.scene_blocks import WgradTable
dev = torch.device("cuda")
def run(M, n_out, k_in, dy, x):
    dW = torch.zeros(n_out, k_in, device=dev); db = torch.zeros(n_out, device=dev)
    t = WgradTable(dev)
    t.add(dy.data_ptr(), dy.stride(0), n_out, x.data_ptr(), x.stride(0), k_in, M, dW.data_ptr(), k_in, db.data_ptr())
    t.launch(_lib.current_stream_ptr(dev)); torch.cuda.synchronize()
    return dW, db
M, n_out, k_in = 32, 128, 128
for (m0, n0, k0) in [(0, 0, 0), (1, 0, 0), (9, 5, 3), (17, 20, 70), (31, 127, 127)]:
    dy = torch.zeros(M, n_out, device=dev); x = torch.zeros(M, k_in, device=dev)
    dy[m0, n0] = 1.0; x[m0, k0] = 1.0
    dW, db = run(M, n_out, k_in, dy, x)
    nz = dW.nonzero().tolist()
    print((m0, n0, k0), "->", nz[:6], float(dW.sum()), "db nz", db.nonzero().flatten().tolist()[:4])
torch.manual_seed(0)
for (M, n_out, k_in) in [(32, 128, 128), (64, 128, 128), (960, 256, 256)]:
    dy = torch.randn(M, n_out, device=dev); x = torch.randn(M, k_in, device=dev)
    dW, db = run(M, n_out, k_in, dy, x)
    want = dy.double().t() @ x.double()
    print(M, n_out, k_in, float((dW.double() - want).norm() / want.norm()), float((db.double() - dy.double().sum(0)).norm()))
