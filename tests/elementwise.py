"""Element-wise float comparison used by every GPU parity test that touches a pinned golden or the oracle.

Rule (north_star: 1e-4 fp32):  |got - ref| <= 1e-4 + 1e-4 |ref|  for EVERY element -- no scaling by the tensor's
maximum.  fp32 caveat, stated rather than hidden: a gradient element is a sum of ~10^4 products; the reference
(torch CPU fp32) and the kernel associate that sum differently, so an element that is a cancellation residue may differ
by more than 1e-4 of the ELEMENT while both sit within fp32 round-off of the exact value.  For an element that misses
the strict test the same oracle evaluated in float64 is the arbiter: the kernel must be within 1e-4 + 1e-4 |ref64| of
it, or at least as close to it as the fp32 reference itself is (x4) -- never looser than what fp32 arithmetic of the
reference can resolve.  The number of arbitrated elements is ASSERTED to stay below 0.1 % (assert_arbiter_rate).
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import cpu_ref

TOL = 1e-4
ARBITER_MAX_FRACTION = 1e-3


def new_stats():
    return {"elements": 0, "arbiter": 0}


def oracle64(net_name, params, batch_cpu, target=None, task="reg", trace=None, **fw):
    """(pred, loss, grads) of oracle/cpu_ref.py evaluated in float64 on the same inputs (the arbiter)."""
    p64 = {k: v.double() for k, v in params.items()}
    b64 = batch_cpu.clone()
    for key in ("x", "edge_attr", "pos", "y", "internal_edge_attr"):
        v = getattr(b64, key, None)
        if torch.is_tensor(v) and v.is_floating_point():
            setattr(b64, key, v.double())
    tgt = b64.y if target is None else target
    if torch.is_tensor(tgt) and tgt.is_floating_point():
        tgt = tgt.double()
    if trace is not None:
        fw = dict(fw, trace=trace)
    return cpu_ref.loss_and_grads(net_name, p64, b64, tgt, task=task, **fw)


def check(name, got, ref32, ref64_fn, stats):
    """got vs the fp32 reference element by element; ref64_fn() -> the float64 arbiter (evaluated lazily)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref32, dtype=np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    nan_mismatch = np.isnan(got) != np.isnan(ref)
    assert not nan_mismatch.any(), "%s: NaN pattern differs at %s" % (name, np.argwhere(nan_mismatch)[:4].tolist())
    both = ~np.isnan(ref)
    bad = both & (np.abs(got - ref) > TOL + TOL * np.abs(ref))
    stats["elements"] += got.size
    if not bad.any():
        return
    ref64 = np.asarray(ref64_fn(), dtype=np.float64).reshape(ref.shape)
    err_kernel = np.abs(got - ref64)
    err_ref32 = np.abs(ref - ref64)
    ok = (err_kernel <= TOL + TOL * np.abs(ref64)) | (err_kernel <= 4.0 * err_ref32 + 1e-7)
    stats["arbiter"] += int(bad.sum())
    fail = bad & ~ok
    if fail.any():
        worst = int(np.argmax(np.where(fail, err_kernel, 0.0)))
        raise AssertionError("%s: element %d got %.9g, fp32 reference %.9g, fp64 oracle %.9g (%d of %d elements fail)" %
                             (name, worst, got.flat[worst], ref.flat[worst], ref64.flat[worst], int(fail.sum()), got.size))


def assert_arbiter_rate(stats, what=""):
    limit = max(1, int(ARBITER_MAX_FRACTION * stats["elements"]))
    assert stats["arbiter"] <= limit, ("%s: %d of %d elements needed the float64 arbiter (limit %d = 0.1 %%)" %
                                       (what, stats["arbiter"], stats["elements"], limit))


class Lazy64:
    """Evaluates the float64 oracle once, on first use."""

    def __init__(self, net_name, params, batch_cpu, target=None, task="reg", want_trace=False, **fw):
        self.args = (net_name, params, batch_cpu, target, task)
        self.fw = fw
        self.want_trace = want_trace
        self.trace = {} if want_trace else None
        self.val = None

    def get(self):
        if self.val is None:
            n, p, b, t, task = self.args
            self.val = oracle64(n, p, b, target=t, task=task, trace=self.trace, **self.fw)
        return self.val

    def pred(self):
        return self.get()[0].numpy()

    def loss(self):
        return float(self.get()[1])

    def grad(self, k):
        return self.get()[2][k].numpy()

    def traced(self, k):
        self.get()
        return self.trace[k].detach().numpy()


def check_step(where, lazy, loss, pred, grads, ref_loss, ref_pred, ref_grads, stats=None):
    """loss / predictions / every gradient of one training step vs (ref_*) with `lazy` as the arbiter."""
    own = stats is None
    stats = new_stats() if own else stats
    check(where + " loss", float(loss), float(ref_loss), lazy.loss, stats)
    check(where + " pred", np.asarray(pred).reshape(-1), np.asarray(ref_pred).reshape(-1), lambda: lazy.pred().reshape(-1), stats)
    assert set(grads) == set(ref_grads), (sorted(grads), sorted(ref_grads))
    for k in sorted(grads):
        check(where + " grad " + k, grads[k], np.asarray(ref_grads[k]), lambda k=k: lazy.grad(k), stats)
    if own:
        assert_arbiter_rate(stats, where)
    return stats
