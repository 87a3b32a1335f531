"""Unit test of the register-resident MFMA chain layer (csrc/mfma_chain.h, host packing pack_chain) through its C-ABI hook:
the operand permutation (identity weights return the input exactly, a single weight moves exactly one feature), the bias path,
and a random layer against a float64 product.  v_mfma_f32_32x32x2_f32 is an exact float32 fma chain in k order, so structured
cases are bit-exact and the random one agrees to float32 accumulation error."""
import numpy as np
import pytest
import torch

from dgdm_amd import _lib

pytestmark = pytest.mark.gpu


def _run(W, b, Xd):
    Yd = torch.zeros_like(Xd)
    W = np.ascontiguousarray(W, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    _lib.check(_lib.lib().dgdm_debug_chain_layer(W.ctypes.data, b.ctypes.data, Xd.data_ptr(), Yd.data_ptr(), None))
    return Yd.cpu().numpy()


def test_chain_layer_structure_and_values():
    assert torch.cuda.is_available()
    _lib.device_init(0)
    rs = np.random.RandomState(0)
    X = rs.randn(32, 256).astype(np.float32)
    Xd = torch.from_numpy(X).to("cuda:0")
    assert np.array_equal(_run(np.eye(256), np.zeros(256), Xd), X)                           # operand layout round trip
    assert np.array_equal(_run(np.zeros((256, 256)), np.arange(256), Xd), np.tile(np.arange(256, dtype=np.float32), (32, 1)))
    for (o, i) in ((3, 7), (255, 0), (64, 200), (31, 32)):                                  # W[o][i] = 1: Y[:, o] = X[:, i], rest 0
        W = np.zeros((256, 256)); W[o, i] = 1.0
        Y = _run(W, np.zeros(256), Xd)
        want = np.zeros_like(X); want[:, o] = X[:, i]
        assert np.array_equal(Y, want), (o, i)
    W, b = rs.randn(256, 256) / 16, rs.randn(256)
    Y = _run(W, b, Xd)
    ref = X.astype(np.float64) @ W.astype(np.float32).astype(np.float64).T + b.astype(np.float32)
    assert np.linalg.norm(Y - ref) / np.linalg.norm(ref) < 1e-6
    # exactly a float32 fma chain starting from the bias, in the chain's K order: block o, accumulator register r, half h
    # -> input feature 32 o + (r & 3) + 8 (r >> 2) + 4 h  (mfma_chain.h)
    acc = np.tile(b.astype(np.float32), (32, 1))
    Wf = W.astype(np.float32)
    for o in range(8):
        for r in range(16):
            for h in range(2):
                k = 32 * o + (r & 3) + 8 * (r >> 2) + 4 * h
                acc = (acc.astype(np.float64) + X[:, k:k + 1].astype(np.float64) * Wf[None, :, k].astype(np.float64)).astype(np.float32)
    assert np.array_equal(Y, acc)
