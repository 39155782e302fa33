import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import numpy as np
import oracle as orc, ops_ref as ref
from gpu_util import backend, gg
L = gg.base()
rng = np.random.default_rng(11)
hd, n_head, n_head_kv, kv_size, n_kv, ntok = 128, 32, 8, 256, 96, 1
kc = rng.uniform(-1, 1, size=(1, 1, kv_size, hd * n_head_kv)).astype(np.float16)
q_ = rng.uniform(-1, 1, size=(1, ntok, n_head, hd)).astype(np.float32)
with gg.Context() as ctx:
    k_l = ctx.new_tensor(gg.F16, (hd * n_head_kv, kv_size)); q_cur = ctx.new_tensor(gg.F32, (hd, n_head, ntok))
    k = L.ggml_view_3d(ctx.ctx, k_l, hd, n_kv, n_head_kv, hd * n_head_kv * 2, hd * 2, 0)
    q = L.ggml_permute(ctx.ctx, q_cur, 0, 2, 1, 3)
    kq = L.ggml_mul_mat(ctx.ctx, k, q)
    be = backend(); ctx.alloc(be)
    for nm, t in (("k_l", k_l), ("k", k), ("q_cur", q_cur), ("q", q), ("kq", kq)):
        c = t.contents
        print(nm, "ne", list(c.ne), "nb", list(c.nb), "data", hex(c.data or 0), "vsrc", bool(c.view_src), "voffs", c.view_offs)
    gg.tensor_set(k_l, kc); gg.tensor_set(q_cur, q_)
    be.compute(gg.graph_of(ctx, kq))
    got = gg.tensor_get(kq)
K = kc[0, 0, :n_kv].reshape(n_kv, n_head_kv, hd).transpose(1, 0, 2)[None].astype(np.float32)
Q = q_.transpose(0, 2, 1, 3)
exp = ref.mul_mat_dense(K, Q)
print("nmse gqa", orc.nmse(exp, got))
# candidates
Kf = kc[0, 0].astype(np.float32)
for h in range(3):
    print("head", h, "got", got[0, h, 0, :3], "exp", exp[0, h, 0, :3])
alt = np.stack([Q[0, h] @ K[0, h % 8].T for h in range(32)])[None]
print("nmse h%8", orc.nmse(alt, got))
