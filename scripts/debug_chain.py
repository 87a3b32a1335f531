import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dgdm_amd import _lib
_lib.device_init(0)
L = _lib.lib()
L.dgdm_debug_chain_layer.restype = C.c_int
L.dgdm_debug_chain_layer.argtypes = [C.c_void_p] * 5
dev = torch.device("cuda:0")
rs = np.random.RandomState(0)
X = torch.from_numpy(rs.randn(32, 256).astype(np.float32))
Xd = X.to(dev); Yd = torch.zeros_like(Xd)
def run(W, b):
    W = np.ascontiguousarray(W.astype(np.float32)); b = np.ascontiguousarray(b.astype(np.float32))
    _lib.check(L.dgdm_debug_chain_layer(W.ctypes.data, b.ctypes.data, Xd.data_ptr(), Yd.data_ptr(), None))
    return Yd.cpu().numpy()
Y = run(np.eye(256), np.zeros(256))
print("identity exact:", np.array_equal(Y, X.numpy()))
if not np.array_equal(Y, X.numpy()):
    # find permutation: for row 0, where does each output come from
    src = []
    for f in range(256):
        m = np.where(np.isclose(X.numpy()[0], Y[0, f]))[0]
        src.append(int(m[0]) if len(m) else -1)
    print("out feature f <- in feature:", src[:64])
    rows_ok = [np.allclose(Y[n], X.numpy()[n]) for n in range(32)]
    print("rows ok", rows_ok)
W = rs.randn(256, 256) / 16; b = rs.randn(256)
Y = run(W, b)
ref = X.numpy().astype(np.float64) @ W.T + b
print("random layer rel err", np.linalg.norm(Y - ref) / np.linalg.norm(ref))
Y = run(np.eye(256), np.zeros(256))
Xn = X.numpy()
print("X[0,:8]", Xn[0,:8]); print("Y[0,:8]", Y[0,:8])
for f in range(6):
    hits = np.argwhere(np.isclose(Xn, Y[0, f], rtol=1e-6, atol=1e-7))
    print("Y[0,%d]=%g found at" % (f, Y[0,f]), hits[:4].tolist())
# zero weights + bias only
Yb = run(np.zeros((256,256)), np.arange(256))
print("bias only row0[:40]", Yb[0,:40])
print("bias only ok:", np.array_equal(Yb, np.tile(np.arange(256, dtype=np.float32), (32,1))))
# single one at W[3][7]
W1 = np.zeros((256,256)); W1[3,7] = 1
Y1 = run(W1, np.zeros(256))
nz = np.argwhere(Y1 != 0)
print("W[3][7]=1: nonzeros", nz[:10].tolist(), "expected Y[n][3] = X[n][7]; Y1[0,3]=", Y1[0,3], "X[0,7]=", Xn[0,7])
if len(nz):
    n0, f0 = nz[0]
    print("value", Y1[n0, f0], "matches X at", np.argwhere(np.isclose(Xn, Y1[n0, f0]))[:3].tolist())
