"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, exports every symbol
include/ppcr.h declares, fails loudly without a GPU, and its host-side closed-form solver agrees
with the oracle.  No compute kernels are launched here."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import binding as po
from probabilistic_point_clouds_registration_amd import _lib, build, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "ppcr.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(ppcr_[a-z_0-9]+)\s*\(", header)))
    assert len(declared) >= 25
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/ppcr.h but not exported"
    assert sorted(_lib.SYMBOLS) == declared
    assert lib.ppcr_abi_version() == 1


def test_documented_options_are_the_implemented_ones():
    """The keys ppcr_set_option accepts (csrc/ppcr_hip_api.inc) and the keys include/ppcr.h documents are the same set: a
    knob that leaves the implementation leaves the header, a new one is described there."""
    header = open(os.path.join(ROOT, "include", "ppcr.h")).read()
    block = header[header.index("Tuning / debugging knobs"):header.index("int ppcr_set_option(")]
    documented = set()
    for line in block.splitlines():
        m = re.match(r'\s*\*\s+((?:"[a-z_0-9]+"(?:,\s*)?)+)\s', line)      # the key(s) a description starts with
        if m:
            documented.update(re.findall(r'"([a-z_0-9]+)"', m.group(1)))
    api = open(os.path.join(ROOT, "probabilistic_point_clouds_registration_amd", "csrc", "ppcr_hip_api.inc")).read()
    body = api[api.index("int ppcr_set_option("):]
    body = body[:body.index("\n}\n")]
    implemented = set(re.findall(r'std::strcmp\(key, "([a-z_0-9]+)"\)', body))
    assert len(implemented) >= 15
    assert documented == implemented, (sorted(documented - implemented), sorted(implemented - documented))


def test_no_silent_cpu_fallback(lib):
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(_lib.PpcrError) as e:
        _lib.Context(0)
    assert e.value.code == -5 and "no CPU fallback" in str(e.value)


def test_missing_extension_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libppcr_hip.so")
    with pytest.raises(ImportError):
        _lib.load()


def test_product_never_imports_oracle():
    """The shipped package must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "probabilistic_point_clouds_registration_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".inc", ".h", ".hpp", ".cpp", ".cc")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                for needle in ("import oracle", "from oracle", "libppcr_oracle", "ppcr_oracle.c", "po_radius_search"):
                    assert needle not in txt, (f, needle)


def test_host_solver_matches_oracle(lib):
    rng = np.random.default_rng(1)
    for trial in range(30):
        n = 40
        x = rng.normal(size=(n, 3)) * rng.uniform(0.5, 5) + rng.normal(size=3) * 20
        Rg = synth.rodrigues(rng.normal(size=3), rng.uniform(0, 3.1))
        y = x @ Rg.T + rng.normal(size=3) + rng.normal(size=(n, 3)) * 0.02
        if trial % 5 == 0:
            x[:, 2] = 1.0                      # planar source: rank-2 cross-covariance
            y = x @ Rg.T
        w = rng.uniform(0.1, 1, size=n)
        c = x.mean(0) + rng.normal(size=3)
        xc, yc = x - c, y - c
        S = np.concatenate([[w.sum()], (w[:, None] * xc).sum(0), (w[:, None] * yc).sum(0),
                            np.einsum("n,na,nb->ab", w, xc, yc).reshape(9), [(w * ((y - x) ** 2).sum(1)).sum()],
                            [(w * (xc ** 2).sum(1)).sum()], [(w * (yc ** 2).sum(1)).sum()]])
        R, t, rc = _lib.solve_moments(S, c)
        Ro, to, rco = po.kabsch(S, c)
        assert rc == rco == 0
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and np.linalg.det(R) > 0
        assert synth.rotation_angle(R, Ro) < 1e-10 and np.linalg.norm(t - to) < 1e-9
        assert abs(_lib.cost_from_moments(S, c, R, t) - po.cost_from_sums(S, c, Ro, to)) < 1e-12 * (S[17] + S[18])
    R, t, rc = _lib.solve_moments(np.zeros(19), np.zeros(3))
    assert rc == 1 and np.allclose(R, np.eye(3)) and np.allclose(t, 0)


def test_stop_rule_is_the_references_has_converged(lib):
    """ppcr_stop_rule_check (one definition for ppcr_align, the C++ class and this binding) against a line-by-line
    model of hasConverged (src/prob_point_cloud_registration.cc:138-158) on random cost-drop sequences, including
    NaN drops, negative n_iter (no cap) and fractional n_cost_drop_it (a double in the params struct)."""
    rng = np.random.default_rng(7)

    class Model:  # the reference's members and control flow
        def __init__(self):
            self.current_iteration, self.cost_drop, self.num_unuseful = 0, 0.0, 0

        def has_converged(self, n_iter, thresh, n_it):
            if self.current_iteration == n_iter:
                return 1
            if self.cost_drop < thresh:
                if self.num_unuseful > n_it:
                    return 2
                self.num_unuseful += 1
            else:
                self.num_unuseful = 0
            return 0

    for trial in range(300):
        n_iter = int(rng.integers(-1, 40))
        thresh = float(rng.choice([0.0, 0.01, 0.5, 2.0]))
        n_it = float(rng.choice([0, 1, 2.5, 5]))
        rule, model = _lib.StopRule(), Model()
        for step in range(60):
            a, b = rule.check(n_iter, thresh, n_it), model.has_converged(n_iter, thresh, n_it)
            assert a == b, (trial, step, a, b)
            if a:
                break
            drop = float(rng.choice([rng.uniform(-0.2, 1.0), 0.0, float("nan"), 1e-3]))
            rule.cost_drop = model.cost_drop = drop
            rule.iteration += 1
            model.current_iteration += 1
        assert rule.iteration == model.current_iteration and rule.idle == model.num_unuseful
    # the default thresholds cannot stop before six iterations (cost_drop starts at 0: the first check is idle already)
    rule = _lib.StopRule()
    n = 0
    while rule.check(1000, 0.01, 5) == 0:
        rule.cost_drop = 0.0
        rule.iteration += 1
        n += 1
    assert n == 6
