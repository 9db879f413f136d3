"""The algebra behind CU2REC_SGD_BLOCKSOLVE (cu2rec_amd/csrc/blocksolve.hip), checked on the host against the
sequential chain of mf_sequential.cu:102-143 restated in numpy.  No GPU, no library: this pins the FORMULAS the
kernels implement ("Block-solve mode").

Inside one iteration every user appears once, so the updates that hit one item y form a chain over the users
x_0 < x_1 < ... that sampled it.  With a = 1 - lr*Q_reg, c = 1 - lr*item_bias_reg and link k of a block of B links
    e_k       = r_k - gb - ub_k - b^(k) - p_k . q^(k)
    q^(k+1)   = a q^(k) + lr e_k p_k              b^(k+1) = c b^(k) + lr e_k
    p_k'      = (1 - lr*P_reg) p_k + lr e_k q^(k)  ub_k'   = (1 - lr*user_bias_reg) ub_k + lr e_k
the errors of the block solve the unit lower triangular system (I + lr L) e = rhs with
    L_kj  = c^(k-1-j) + a^(k-1-j) (p_k . p_j)   (j < k)           -- phase 1: a Gram matrix, no chain state in it
    rhs_k = (r_k - gb - ub_k) - c^k b^(0) - a^k (p_k . q^(0))      -- phase 2: one mat-vec with the block's rows
and the block leaves q^(B) = a^B q^(0) + lr sum_j a^(B-1-j) e_j p_j, b^(B) likewise.                -- phase 2
The user rows follow from q^(k) = a^k q^(0) + lr sum_{j<k} a^(k-1-j) e_j p_j.                       -- phase 3
"""
import numpy as np
import pytest


def sequential_chain(P, ub, r, q, b, gb, lr, regs, dtype):
    """mf_sequential.cu:119-141 for the users of one item, in order."""
    p_reg, q_reg, ub_reg, ib_reg = (dtype(v) for v in regs)
    lr = dtype(lr)
    P, ub, q, b = P.astype(dtype).copy(), ub.astype(dtype).copy(), q.astype(dtype).copy(), dtype(b)
    errs = np.zeros(len(r), dtype)
    for k in range(len(r)):
        pred = dtype(gb) + ub[k] + b + np.dot(P[k], q).astype(dtype)
        e = dtype(r[k]) - pred
        p_old, q_old = P[k].copy(), q.copy()
        P[k] = p_old + lr * (e * q_old - p_reg * p_old)
        q = q_old + lr * (e * p_old - q_reg * q_old)
        ub[k] = ub[k] + lr * (e - ub_reg * ub[k])
        b = b + lr * (e - ib_reg * b)
        errs[k] = e
    return P, ub, q, b, errs


def blocked_chain(P, ub, r, q, b, gb, lr, regs, B, dtype):
    """The three phases of the block-solve mode, block by block.  Decay factors are carried as 1 - delta with delta
    rounded to `dtype` (x * a^k == x - delta_k * x): rounding a^k itself would bias the decay rate of a hot item by
    up to one ulp of 1.0 per block, the same systematic error every block."""
    p_reg, q_reg, ub_reg, ib_reg = (dtype(v) for v in regs)
    lr = dtype(lr)
    a, c = 1.0 - float(lr) * float(q_reg), 1.0 - float(lr) * float(ib_reg)  # double, from the rounded constants
    ap, au = dtype(1.0 - float(lr) * float(p_reg)), dtype(1.0 - float(lr) * float(ub_reg))
    P0, ub0 = P.astype(dtype), ub.astype(dtype)
    q, b = q.astype(dtype).copy(), dtype(b)
    P_new, ub_new = np.empty_like(P0), np.empty_like(ub0)
    errs = np.zeros(len(r), dtype)
    apow = (a ** np.arange(B + 1)).astype(dtype)          # only ever multiplied into lr-sized terms
    cpow = (c ** np.arange(B + 1)).astype(dtype)
    adel = (1.0 - a ** np.arange(B + 1)).astype(dtype)    # x * a^k  ==  x - adel[k] * x
    cdel = (1.0 - c ** np.arange(B + 1)).astype(dtype)
    for m0 in range(0, len(r), B):
        Pm, n = P0[m0:m0 + B], min(B, len(r) - m0)
        # phase 1: Gram matrix -> lr * L (strictly lower), independent of q and b
        G = (Pm @ Pm.T).astype(dtype)
        k, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
        low = j < k
        d = np.where(low, k - 1 - j, 0)
        L = np.where(low, lr * (cpow[d] + apow[d] * G), dtype(0)).astype(dtype)
        # phase 2: rhs from the block's start state, forward substitution (column sweep), end state
        base = (r[m0:m0 + n].astype(dtype) - dtype(gb)) - ub0[m0:m0 + n]
        dots = (Pm @ q).astype(dtype)
        rhs = ((base - (b - cdel[:n] * b)) - (dots - adel[:n] * dots)).astype(dtype)
        e = rhs.copy()
        for jj in range(n):
            e[jj + 1:] -= L[jj + 1:, jj] * e[jj]
        w = (lr * apow[n - 1 - np.arange(n)] * e).astype(dtype)
        q_start, q = q, ((q - adel[n] * q) + w @ Pm).astype(dtype)
        b = (b - cdel[n] * b) + np.sum(lr * cpow[n - 1 - np.arange(n)] * e, dtype=dtype)
        # phase 3: item row as each link saw it -> user rows
        T = np.where(low, lr * apow[d] * e[None, :], dtype(0)).astype(dtype)
        Qh = ((q_start[None, :] - adel[:n, None] * q_start[None, :]) + T @ Pm).astype(dtype)
        P_new[m0:m0 + n] = ap * Pm + lr * e[:, None] * Qh
        ub_new[m0:m0 + n] = au * ub0[m0:m0 + n] + lr * e
        errs[m0:m0 + n] = e
    return P_new, ub_new, q, b, errs


@pytest.mark.parametrize("n,f,B", [(1, 8, 32), (31, 10, 32), (32, 100, 32), (33, 100, 32), (200, 100, 64),
                                   (1979, 100, 64), (500, 128, 32)])
def test_blocked_chain_equals_sequential_f64(n, f, B):
    rng = np.random.default_rng(n * 1000 + f)
    P, q = rng.normal(0, 0.3, (n, f)), rng.normal(0, 0.3, f)
    ub, b, r = rng.normal(0, 0.3, n), 0.2, rng.integers(1, 11, n) / 2.0
    regs = (0.02, 0.03, 0.04, 0.05)
    want = sequential_chain(P, ub, r, q, b, 3.5, 0.01, regs, np.float64)
    got = blocked_chain(P, ub, r, q, b, 3.5, 0.01, regs, B, np.float64)
    for g, w in zip(got, want):
        assert np.allclose(g, w, rtol=0, atol=1e-11), np.abs(np.asarray(g) - np.asarray(w)).max()


def test_blocked_chain_f32_rounding_gap_is_small():
    """In float32 the two orders differ only by rounding: far inside the 1e-4 RMSE contract."""
    rng = np.random.default_rng(7)
    n, f = 1979, 100  # the hottest item of the ML-20M-shape set receives about this many updates per iteration
    P, q = rng.normal(0, 0.3, (n, f)), rng.normal(0, 0.3, f)
    ub, b, r = rng.normal(0, 0.3, n), 0.2, rng.integers(1, 11, n) / 2.0
    regs = (0.02, 0.02, 0.02, 0.02)
    want = sequential_chain(P, ub, r, q, b, 3.5, 0.01, regs, np.float32)
    for B in (32, 64):
        got = blocked_chain(P, ub, r, q, b, 3.5, 0.01, regs, B, np.float32)
        for g, w in zip(got, want):
            assert np.abs(np.asarray(g, np.float64) - np.asarray(w, np.float64)).max() < 2e-5


def lookahead_chain(P, ub, r, q, b, gb, lr, regs, dtype):
    """The look-ahead form of the chain (bs_chain_kernel, round 4).  Blocks of B = 64 links; block i >= 1 takes its
    right-hand side from the state in front of block i - 1 and the errors of block i - 1:
        e_i   = M_i (pre_i - N_i e_(i-1)),   pre_i[k] = base_i[k] - c^(k+64) b_(i-1) - a^(k+64) (p_(i,k) . q_(i-1))
        N_i[k][j] = lr (c^(63+k-j) + a^(63+k-j) (p_(i,k) . p_(i-1,j)))            -- the block of lr L below the diagonal block
    (M_i = (I + N_ii)^-1 as before; block 0 as in blocked_chain).  N_i is a cross Gram matrix of two neighbouring blocks' rows --
    no chain state in it, phase 1 builds it -- so the only dependent work per block is two 64 x 64 mat-vecs; the state
    (q_i, b_i) follows the errors one block behind (q_(i+1) = a^n q_i + lr sum_j a^(n-1-j) e_i[j] p_(i,j)) and is needed
    again only for pre_(i+2)."""
    B = 64
    p_reg, q_reg, ub_reg, ib_reg = (dtype(v) for v in regs)
    lr = dtype(lr)
    a, c = 1.0 - float(lr) * float(q_reg), 1.0 - float(lr) * float(ib_reg)
    ap, au = dtype(1.0 - float(lr) * float(p_reg)), dtype(1.0 - float(lr) * float(ub_reg))
    P0, ub0 = P.astype(dtype), ub.astype(dtype)
    apow = (a ** np.arange(2 * B + 1)).astype(dtype)
    cpow = (c ** np.arange(2 * B + 1)).astype(dtype)
    adel = (1.0 - a ** np.arange(2 * B + 1)).astype(dtype)
    cdel = (1.0 - c ** np.arange(2 * B + 1)).astype(dtype)
    nblk = (len(r) + B - 1) // B
    qs, bs = [q.astype(dtype).copy()], [dtype(b)]  # state in front of every block
    errs = np.zeros(len(r), dtype)
    e_prev = None
    for i in range(nblk):
        m0 = i * B
        Pm, n = P0[m0:m0 + B], min(B, len(r) - m0)
        kk = np.arange(n)
        k, j = np.meshgrid(kk, kk, indexing="ij")
        low = j < k
        d = np.where(low, k - 1 - j, 0)
        G = (Pm @ Pm.T).astype(dtype)
        Nd = np.where(low, lr * (cpow[d] + apow[d] * G), dtype(0)).astype(dtype)
        M = np.linalg.inv((np.eye(n) + Nd).astype(np.float64)).astype(dtype)
        base = (r[m0:m0 + n].astype(dtype) - dtype(gb)) - ub0[m0:m0 + n]
        if i == 0:
            dots = (Pm @ qs[0]).astype(dtype)
            t = ((base - (bs[0] - cdel[kk] * bs[0])) - (dots - adel[kk] * dots)).astype(dtype)
        else:
            Pp = P0[m0 - B:m0]
            X = (Pm @ Pp.T).astype(dtype)
            kx, jx = np.meshgrid(kk, np.arange(B), indexing="ij")
            dx = B - 1 + kx - jx
            N = (lr * (cpow[dx] + apow[dx] * X)).astype(dtype)
            g = (Pm @ qs[i - 1]).astype(dtype)
            pre = ((base - (bs[i - 1] - cdel[kk + B] * bs[i - 1])) - (g - adel[kk + B] * g)).astype(dtype)
            t = (pre - N @ e_prev).astype(dtype)
        e = (M @ t).astype(dtype)
        errs[m0:m0 + n] = e
        w = (lr * apow[n - 1 - kk] * e).astype(dtype)
        qs.append(((qs[i] - adel[n] * qs[i]) + w @ Pm).astype(dtype))
        bs.append((bs[i] - cdel[n] * bs[i]) + np.sum(lr * cpow[n - 1 - kk] * e, dtype=dtype))
        e_prev = e
    # the user side as in blocked_chain (phase 3), from the start states and the errors
    P_new, ub_new = np.empty_like(P0), np.empty_like(ub0)
    for i in range(nblk):
        m0 = i * B
        Pm, n = P0[m0:m0 + B], min(B, len(r) - m0)
        kk = np.arange(n)
        k, j = np.meshgrid(kk, kk, indexing="ij")
        low = j < k
        d = np.where(low, k - 1 - j, 0)
        e = errs[m0:m0 + n]
        T = np.where(low, lr * apow[d] * e[None, :], dtype(0)).astype(dtype)
        Qh = ((qs[i][None, :] - adel[:n, None] * qs[i][None, :]) + T @ Pm).astype(dtype)
        P_new[m0:m0 + n] = ap * Pm + lr * e[:, None] * Qh
        ub_new[m0:m0 + n] = au * ub0[m0:m0 + n] + lr * e
    return P_new, ub_new, qs[-1], bs[-1], errs


@pytest.mark.parametrize("n,f", [(1, 8), (64, 10), (65, 10), (127, 100), (200, 100), (1979, 100), (450, 112)])
def test_lookahead_chain_equals_sequential_f64(n, f):
    rng = np.random.default_rng(n * 1000 + f + 2)
    P, q = rng.normal(0, 0.3, (n, f)), rng.normal(0, 0.3, f)
    ub, b, r = rng.normal(0, 0.3, n), 0.2, rng.integers(1, 11, n) / 2.0
    regs = (0.02, 0.03, 0.04, 0.05)
    want = sequential_chain(P, ub, r, q, b, 3.5, 0.01, regs, np.float64)
    got = lookahead_chain(P, ub, r, q, b, 3.5, 0.01, regs, np.float64)
    for g, w in zip(got, want):
        assert np.allclose(g, w, rtol=0, atol=1e-11), np.abs(np.asarray(g) - np.asarray(w)).max()


def test_lookahead_chain_f32_rounding_gap_is_small():
    rng = np.random.default_rng(9)
    n, f = 1979, 100
    P, q = rng.normal(0, 0.3, (n, f)), rng.normal(0, 0.3, f)
    ub, b, r = rng.normal(0, 0.3, n), 0.2, rng.integers(1, 11, n) / 2.0
    regs = (0.02, 0.02, 0.02, 0.02)
    want = sequential_chain(P, ub, r, q, b, 3.5, 0.01, regs, np.float32)
    got = lookahead_chain(P, ub, r, q, b, 3.5, 0.01, regs, np.float32)
    for g, w in zip(got, want):
        assert np.abs(np.asarray(g, np.float64) - np.asarray(w, np.float64)).max() < 2e-5
