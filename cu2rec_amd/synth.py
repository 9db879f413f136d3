"""Synthetic rating sets with the shape of the reference's datasets (no network: the real
MovieLens / Netflix files are not available).  Deterministic in `seed`.

Generator: per-user degree ~ log-normal clipped to [min_degree, n_items],
rescaled to the target nnz; a user's items are drawn from a Zipf(alpha) popularity law over a
fixed random permutation of the items (duplicates dropped, so the realised nnz is a little
below the target); rating = clip(round_to_half(3.5 + b_u + b_i + p*_u . q*_i + noise), 0.5, 5)
with planted rank-8 factors, so RMSE has a meaningful floor (about the noise sigma).  An 80/20
split by rating, both halves sorted by user like preprocessing/split_to_test_train.py:39-49.
"""
import numpy as np

from .api import HostCSR

SHAPES = {
    # name: (users, items, nnz, min_degree)        SURVEY.md section 8 table
    "ml-100k": (943, 1682, 100_000, 20),
    "ml-1m": (6040, 3706, 1_000_209, 20),
    "ml-20m": (138_493, 26_744, 20_000_263, 20),
    "netflix": (480_189, 17_770, 100_480_507, 1),
    # not a reference experiment: the next MovieLens size up, 20 user rows per group (4 of them in LDS) in resident launches
    "ml-25m": (162_541, 59_047, 25_000_095, 20),
}


def _csr(user, item, rating, rows, cols):
    counts = np.bincount(user, minlength=rows)
    indptr = np.zeros(rows + 1, np.int32)
    np.cumsum(counts, out=indptr[1:])
    gb = float(np.float32(rating.astype(np.float64).sum() / max(len(rating), 1)))
    return HostCSR(indptr, item.astype(np.int32), rating.astype(np.float32), rows, cols, gb)


def make_ratings(n_users, n_items, nnz, min_degree=20, seed=20240917, zipf_alpha=1.0, sigma=1.0, rank=8,
                 noise=0.8, test_fraction=0.2, integer_ratings=False):
    """-> (train HostCSR, test HostCSR).  Both have rows = n_users, cols = n_items."""
    rng = np.random.default_rng(seed)
    # degrees
    deg = np.exp(rng.normal(0.0, sigma, n_users))
    deg = deg * (nnz / deg.sum())
    deg = np.clip(np.rint(deg), min_degree, n_items).astype(np.int64)
    for _ in range(4):  # rescale so the sum lands near nnz despite the clipping
        free = deg > min_degree
        excess = deg.sum() - nnz
        if abs(excess) < 0.001 * nnz or not free.any():
            break
        deg[free] = np.clip(np.rint(deg[free] * (1.0 - excess / deg[free].sum())), min_degree, n_items)
    # items: Zipf over a random permutation, inverse-CDF sampling, oversample then drop duplicates
    weights = 1.0 / np.power(np.arange(1, n_items + 1, dtype=np.float64), zipf_alpha)
    cdf = np.cumsum(weights)
    cdf /= cdf[-1]
    perm = rng.permutation(n_items)
    key = np.zeros(0, np.int64)
    need = deg.copy()
    for _ in range(4):  # top-up rounds: Zipf draws collide often for heavy users
        over = np.minimum(np.ceil(need * 1.3).astype(np.int64) + (need > 0) * 2, 4 * n_items)
        u_new = np.repeat(np.arange(n_users, dtype=np.int64), over)
        i_new = perm[np.searchsorted(cdf, rng.random(u_new.shape[0]), side="right").clip(0, n_items - 1)]
        key = np.unique(np.concatenate([key, u_new * n_items + i_new]))  # sorted by user, then item; no duplicates
        have = np.bincount(key // n_items, minlength=n_users)
        need = np.maximum(deg - have, 0)
        if need.sum() < 0.002 * nnz:
            break
    user, item = key // n_items, key % n_items
    # trim each user to its target degree (keep a random subset)
    order = rng.random(user.shape[0])
    idx = np.lexsort((order, user))
    user, item = user[idx], item[idx]
    start = np.zeros(n_users + 1, np.int64)
    np.cumsum(np.bincount(user, minlength=n_users), out=start[1:])
    rank_in_user = np.arange(user.shape[0]) - start[user]
    keep = rank_in_user < deg[user]
    user, item = user[keep], item[keep]
    # planted model -> ratings
    pu = rng.normal(0, 0.3, (n_users, rank)).astype(np.float32)
    qi = rng.normal(0, 0.3, (n_items, rank)).astype(np.float32)
    bu = rng.normal(0, 0.4, n_users).astype(np.float32)
    bi = rng.normal(0, 0.4, n_items).astype(np.float32)
    r = 3.5 + bu[user] + bi[item] + np.einsum("ij,ij->i", pu[user], qi[item]) + rng.normal(0, noise, user.shape[0])
    if integer_ratings:
        r = np.clip(np.rint(r), 1.0, 5.0)
    else:
        r = np.clip(np.rint(r * 2.0) / 2.0, 0.5, 5.0)
    r = r.astype(np.float32)
    # every user keeps at least one training rating
    is_test = rng.random(user.shape[0]) < test_fraction
    first = np.zeros(user.shape[0], bool)
    first[np.unique(user, return_index=True)[1]] = True
    is_test &= ~first
    tr, te = ~is_test, is_test
    return (_csr(user[tr], item[tr], r[tr], n_users, n_items), _csr(user[te], item[te], r[te], n_users, n_items))


def make_named(name, seed=20240917, scale=1.0):
    """One of SHAPES, optionally scaled down (users, items and nnz all times `scale`)."""
    users, items, nnz, min_deg = SHAPES[name]
    if scale != 1.0:
        users, items, nnz = max(int(users * scale), 8), max(int(items * scale), 8), max(int(nnz * scale), 64)
    return make_ratings(users, items, nnz, min_degree=min_deg, seed=seed, integer_ratings=(name == "netflix"))


def write_csv(path, csr):
    """`userId,itemId,rating` with header and 1-based ids -- the reference's input format (util.cu:10-16)."""
    user = np.repeat(np.arange(csr.rows), np.diff(csr.indptr)) + 1
    with open(path, "w") as fh:
        fh.write("userId,itemId,rating\n")
        np.savetxt(fh, np.column_stack([user, csr.indices + 1, csr.data]), fmt=["%d", "%d", "%.1f"], delimiter=",")
