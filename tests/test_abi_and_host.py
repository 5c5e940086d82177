"""CPU: the C-ABI library loads, exports every symbol include/fcm.h declares,
fails loudly without a GPU, and its host-side logic (Graph surface, .flag I/O,
Bounds arithmetic) agrees with the oracle.  No kernel runs here."""
import os
import re

import numpy as np
import pytest

from helpers import known_answers, load_flag_fixture

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "fcm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(fcm_[a-z0-9_]+|flagser_count_unweighted)\s*\(", text))
    return sorted(names - {"fcm_node"})  # `const fcm_node (*edges)[2]` is a type, not a function


def test_header_symbols_are_exported(fcm):
    import ctypes
    L = ctypes.CDLL(fcm.LIB_PATH)
    names = _declared_symbols()
    assert "flagser_count_unweighted" in names and len(names) > 30
    for n in names:
        assert hasattr(L, n), "include/fcm.h declares %s but libfcm.so does not export it" % n


def test_python_binding_covers_every_declared_symbol(fcm):
    from flag_complex_mcmc_amd import _ffi
    assert sorted(_ffi.SIGNATURES) == _declared_symbols()


def test_no_gpu_fails_loudly(fcm):
    if fcm.device_count() > 0:
        pytest.skip("a GPU is visible")
    g = fcm.Graph.from_edges(3, [(0, 1), (1, 2)])
    with pytest.raises(fcm.FcmError) as ei:
        g.flagser_count()
    assert ei.value.code == 2 and "no CPU path" in str(ei.value)
    with pytest.raises(fcm.FcmError):
        fcm.MCMCSampler(g, fcm.Bounds([3, 2], [3, 2]), n_chains=1)
    with pytest.raises(fcm.FcmError):
        fcm.count_unweighted(3, [(0, 1)])


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "flag_complex_mcmc_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "oracle_ffi" not in txt and "fcm_oracle" not in txt and "libfcm_oracle" not in txt, os.path.join(dp, f)


def test_graph_surface(fcm, oracle):
    n, e = load_flag_fixture("bug_calc_relax_de.flag")
    g = fcm.Graph.from_edges(n, e)
    o = oracle.Graph.from_edges(n, e)
    assert g.nnodes() == n and g.nedges() == len(o.edges())
    assert (g.edges() == o.edges()).all()
    assert (g.undirected_edges() == o.undirected_edges()).all()
    a, b = map(int, e[0])
    assert g.has_edge(a, b) and not g.has_edge(n + 5, 0)
    g.remove_edge(a, b)
    assert not g.has_edge(a, b) and g.nedges() == len(e) - 1
    g.set_edge(a, b, True)
    g.add_edge(a, b)  # idempotent
    assert g.nedges() == len(e)
    with pytest.raises(fcm.FcmError):
        g.add_edge(0, n)
    h = g.clone()
    h.remove_edge(a, b)
    assert g.has_edge(a, b)
    g2 = fcm.Graph.new_disconnected(5)
    assert g2.nnodes() == 5 and g2.nedges() == 0 and len(g2.undirected_edges()) == 0


def test_flag_file_io(fcm, golden_dir, tmp_path):
    for f in sorted(os.listdir(golden_dir)):
        if not f.endswith(".flag"):
            continue
        n, e = load_flag_fixture(f)
        g = fcm.read_flag_file(os.path.join(golden_dir, f))
        assert g.nnodes() == n
        assert sorted(map(tuple, g.edges().tolist())) == sorted(set(map(tuple, e.tolist())))
        out = tmp_path / f
        fcm.save_flag_file(str(out), g)
        lines = out.read_text().split("\n")
        assert lines[0] == "dim 0:" and lines[2] == "dim 1:"                 # src/io.rs:38-40
        assert lines[1] == " ".join(["1"] * n)
        assert lines[3] == "%d %d 1" % tuple(g.edges()[0])                   # sorted, weight column (io.rs:41-45)
        g2 = fcm.read_flag_file(str(out))
        assert (g2.edges() == g.edges()).all()


def test_flag_file_edge_cases(fcm, tmp_path):
    p = tmp_path / "a.flag"
    p.write_text("dim 0:\n1  1 1 \ndim 1:\n0 1\n\n 1   2 0.5 extra\r\n2\n")
    g = fcm.read_flag_file(str(p))
    assert g.nnodes() == 3 and g.edges().tolist() == [[0, 1], [1, 2]]       # blank / one-token lines skipped (io.rs:30)
    p.write_text("dim 0:\n1 1\ndim 1:\n0 x\n")
    with pytest.raises(fcm.FcmError):                                       # reference: parse().unwrap() panics
        fcm.read_flag_file(str(p))
    with pytest.raises(fcm.FcmError):
        fcm.read_flag_file(str(tmp_path / "missing.flag"))
    p.write_text("dim 0:\n\ndim 1:\n")
    assert fcm.read_flag_file(str(p)).nnodes() == 0                         # empty graph


def test_target_bounds_and_check_match_oracle(fcm, oracle):
    fc = [1000, 100151, 1005709, 1013125, 102824, 1084]
    for r in (0.01, 0.05, 0.0):
        b = fcm.Bounds.target(fc, r)
        mn, mx = oracle.target_bounds(fc, r).lists()
        assert b.flag_count_min == mn and b.flag_count_max == mx
    b = fcm.Bounds([1, 2, 3], [1, 2, 5, 10])
    assert b.check([1, 2, 4]) and b.check([1, 2, 5, 10]) and not b.check([1, 2, 6])
    assert not b.check([1, 2, 4, 0, 1])      # longer than max: padded with 0 (src/util.rs:53-57)
    assert not b.check([1, 2])               # shorter than min: 0 < 3
    assert fcm.default_sample_distance(100151) == oracle.default_sample_distance(100151)


def test_graph_generators_are_deterministic(fcm):
    from flag_complex_mcmc_amd import graphs
    e1 = graphs.random_with_p(200, 0.1, seed=0)
    e2 = graphs.random_with_p(200, 0.1, seed=0)
    assert (e1 == e2).all() and (e1[:, 0] != e1[:, 1]).all()
    assert abs(len(e1) - 0.1 * 200 * 200) < 600
    e3 = graphs.random_edge_draws(1000, 5000, seed=1)
    assert len(np.unique(e3[:, 0].astype(np.int64) * 1000 + e3[:, 1])) == len(e3) and (e3[:, 0] != e3[:, 1]).all()
    ka = known_answers()
    assert graphs.simplex(3).tolist() == ka["ex00"]["edges"]
    assert graphs.clique(3).tolist() == ka["ex03"]["edges"]
    s = graphs.seoify(graphs.clique(4), seed=0)
    assert len(s) == 10


# ---- robustness of the boundary: nothing throws, nothing allocates from unchecked sizes -------
def _state_header(n, n_chains, U, D, ncounts):
    """The fixed part of a libfcm state file with a valid magic (layout of StateHeader in fcm_host.cpp)."""
    import ctypes as C
    from flag_complex_mcmc_amd import _ffi

    class Header(C.Structure):
        _fields_ = [("magic", C.c_char * 8), ("sample_number", C.c_uint64), ("n", C.c_uint32), ("n_chains", C.c_uint32),
                    ("U", C.c_uint64), ("D", C.c_uint64), ("ncounts", C.c_int32), ("reserved", C.c_int32),
                    ("cfg", _ffi.CSamplerConfig), ("bounds", _ffi.CBounds)]
    h = Header()
    h.magic = b"FCMSTAT3"
    h.n, h.n_chains, h.U, h.D, h.ncounts = n, n_chains, U, D, ncounts
    h.cfg.n_chains = n_chains
    h.cfg.move_weights[0] = h.cfg.move_weights[1] = 0.5
    return bytes(h)


@pytest.mark.parametrize("U,D,n_chains,tail", [
    (2 ** 60, 0, 1, 0),          # would be a 2^63-byte vector (std::length_error -> terminate before the fix)
    (3, 2 ** 40, 1, 64),         # D > U
    (3, 1, 2 ** 31, 64),         # chain count the file cannot hold
    (3, 1, 1, 7),                # plausible header, file too short
])
def test_load_state_rejects_corrupt_headers_without_allocating(fcm, tmp_path, U, D, n_chains, tail):
    p = tmp_path / "bad.state"
    p.write_bytes(_state_header(4, n_chains, U, D, 4) + b"\0" * tail)
    with pytest.raises(fcm.FcmError) as ei:
        fcm.MCMCSampler.load_state(str(p))
    assert ei.value.code == 5, str(ei.value)


def test_load_state_rejects_old_magic_and_short_files(fcm, tmp_path):
    p = tmp_path / "old.state"
    p.write_bytes(b"FCMSTAT2" + b"\0" * 600)
    with pytest.raises(fcm.FcmError) as ei:
        fcm.MCMCSampler.load_state(str(p))
    assert ei.value.code == 5
    p.write_bytes(b"FCMSTAT3")
    with pytest.raises(fcm.FcmError):
        fcm.MCMCSampler.load_state(str(p))


def test_default_sample_distance_edge_cases(fcm, oracle):
    # E = 0: ceil(2*0*log2(0)) is NaN, `NaN as usize` = 0 in the reference (src/bin/sample.rs:102); E = 1: log2(1) = 0
    assert fcm.default_sample_distance(0) == 0
    assert fcm.default_sample_distance(1) == 0
    assert fcm.default_sample_distance(2) == 4
    for e in (3, 18, 1961, 100151):
        assert fcm.default_sample_distance(e) == oracle.default_sample_distance(e)


def test_graph_allocation_failure_is_an_error_code(fcm):
    # 200000 vertices: the bitmap would pass 4 GiB of 32-bit offsets -> FCM_ERR_UNSUPPORTED, not an exception
    with pytest.raises(fcm.FcmError) as ei:
        fcm.Graph.new_disconnected(200000)
    assert ei.value.code == 4


def test_multi_wave_kernels_have_no_spill_traffic_in_their_loop():
    """hipcc cross-compiles here: the multi-wave kernels must keep their proposal loop free of scratch (VGPR spill)
    instructions -- at 64 VGPRs a small edit can tip hipcc's allocation, and spill traffic inside the loop costs tens of
    per cent (tools/scratch_census.sh lists every variant).  About a dozen such instructions belong to the
    two out-of-line calls.  Checked here: the headline kernel (m4) and the ones the other BASELINE configs run."""
    import subprocess
    src = os.path.join(ROOT, "flag_complex_mcmc_amd", "csrc")
    procs = {}
    for tag in ("m4", "m6", "n2", "n3", "n4"):
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
               "-DFCM_TAG=%s_0" % tag, "-DFCM_MAXT=%s" % tag[1], "-DFCM_EXACT=1", "-DFCM_PC=%d" % (1 if tag[0] == "m" else 2),
               "-DFCM_CLIQUE=0", "-S", "--cuda-device-only", "-o", "-", "fcm_step_variant.hip"]
        procs[tag] = subprocess.Popen(cmd, cwd=src, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    for tag, pr in procs.items():
        out, err = pr.communicate(timeout=900)
        assert pr.returncode == 0, err[-2000:]
        body = out[out.index("_Z18fcm_step_mw_kernel"):]
        body = body[:body.index("s_endpgm")]
        assert body.count("scratch_") <= 16, (tag, body.count("scratch_"))
