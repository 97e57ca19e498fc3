"""The reference's roco selection as a SET of valid outcomes (torch.topk takes an arbitrary subset of a tied class), plus seeded score
rows with NaN standard deviations.  Used by tests/test_hip_select_edges.py (GPU) and tests/test_select_rule_cpu.py."""
import numpy as np
import torch


# ---------------------------------------------------------------------------------------------------------------------
# the reference's decision as a SET of outcomes
# ---------------------------------------------------------------------------------------------------------------------
def feasible_classes(std_row, k1):
    """std row -> (forced members, tied pool, how many of the pool topk must take).  Keys are ranked the way torch.topk(largest=False)
    does: ascending, NaN last; members equal to the k1-th key (all NaNs count as equal) form the pool."""
    std = std_row.double().numpy()
    nan = np.isnan(std)
    order = np.argsort(np.where(nan, np.inf, std), kind="stable")
    kth = order[k1 - 1]
    if nan[kth]:
        pool = np.nonzero(nan)[0]
        forced = np.nonzero(~nan)[0]
    else:
        tau = std[kth]
        pool = np.nonzero(~nan & (std == tau))[0]
        forced = np.nonzero(~nan & (std < tau))[0]
    return forced, pool, k1 - len(forced)


def valid_victims(victims, mean_row, forced, pool, need, k):
    """Is `victims` (k indices) what `feas[topk(mean[feas], k, smallest)]` returns for SOME admissible feasible set forced + O,
    O a `need`-subset of the pool?  The friendliest O for a given victim set holds the victims' own pool members and otherwise the
    pool members with the largest means."""
    victims = [int(v) for v in victims]
    mean = mean_row.double().numpy()
    fs, ps = set(forced.tolist()), set(pool.tolist())
    if len(set(victims)) != k or any(v not in fs and v not in ps for v in victims):
        return False
    mine = [v for v in victims if v in ps]
    if len(mine) > need:
        return False
    rest = sorted((p for p in ps if p not in victims), key=lambda p: -mean[p])
    chosen = mine + rest[:need - len(mine)]
    if len(chosen) != need:
        return False
    feas = np.array(sorted(fs | set(chosen)))
    kth = np.sort(mean[feas], kind="stable")[k - 1]
    # the victims are the k smallest means of this feasible set: everything strictly below the k-th smallest mean is a victim, and no
    # victim lies above it (equal means at the cut are their own arbitrary tie)
    return all(mean[v] <= kth for v in victims) and all(int(f) in victims for f in feas if mean[f] < kth)


def seed_rows(H, W, n_live, n_nan, g, c_lo=10, c_hi=60, nan_lo=0, nan_hi=None):
    """S, Q, C [H, W]: the first `n_live` columns get a count in [c_lo, c_hi), a mean in [1, 2) and a variance in [0.01, 0.5); `n_nan`
    of the columns [nan_lo, nan_hi) instead get a radicand that stays clearly negative after one more step (NaN std) and the lowest
    means of the row."""
    nan_hi = n_live if nan_hi is None else nan_hi
    s, q, c = torch.zeros(H, W), torch.zeros(H, W), torch.zeros(H, W)
    nan_cols = []
    for h in range(H):
        cc = torch.randint(c_lo, c_hi, (n_live,), generator=g).float()
        mean = 1.0 + torch.rand(n_live, generator=g)
        var = 0.01 + 0.5 * torch.rand(n_live, generator=g)
        sv, qv = mean * cc, (mean * mean + var) * cc
        cols = nan_lo + torch.randperm(nan_hi - nan_lo, generator=g)[:n_nan]
        sv[cols] = 0.5 * cc[cols] * (1.0 + 0.1 * torch.rand(n_nan, generator=g))     # mean ~ 0.5: the lowest means of the row
        qv[cols] = 0.2 * cc[cols]                                                  # Q/C = 0.2 < mean^2 >= 0.25
        s[h, :n_live], q[h, :n_live], c[h, :n_live] = sv, qv, cc
        nan_cols.append(set(cols.tolist()))
    return s, q, c, nan_cols


class Capture:
    """oracle.SELECT_HOOK: keeps the rows the selection saw."""

    def __call__(self, fn, policy, s, q, c, args, ids):
        self.s, self.q, self.c, self.args, self.ids = s.clone(), q.clone(), c.clone(), args, ids.clone()


def _mk(L, H, n, D, g):
    return torch.randn(L, H, n, D, generator=g).half()


