"""The C++ host layer above the C ABI: the drop-in classes (tests/cpp/test_api.cc restates the reference's
gtest cases) and the `probabilistic_point_cloud_registration` CLI with the reference's flags and outputs."""
import os
import re
import subprocess

import numpy as np
import pytest

from oracle import binding as po
from probabilistic_point_clouds_registration_amd import build, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def programs():
    return build.build_host_programs()


def write_pcd(path, pts, binary=False):
    pts = np.asarray(pts, np.float32)[:, :3]
    hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\n"
           f"WIDTH {len(pts)}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {len(pts)}\nDATA {'binary' if binary else 'ascii'}\n")
    with open(path, "wb") as f:
        f.write(hdr.encode())
        if binary:
            f.write(np.ascontiguousarray(pts).tobytes())
        else:
            for p in pts:
                f.write(("%.9g %.9g %.9g\n" % tuple(p)).encode())


def lzf_compress(data):
    """Small greedy LZF encoder for the tests (literal runs + back references with a 3-byte hash), producing the
    stream format liblzf defines; the decoder under test is the C++ one."""
    data = bytes(data)
    out = bytearray()
    lit = bytearray()
    table = {}
    i, n = 0, len(data)

    def flush():
        nonlocal lit
        for k in range(0, len(lit), 32):
            chunk = lit[k:k + 32]
            out.append(len(chunk) - 1)
            out.extend(chunk)
        lit = bytearray()

    while i < n:
        ref = table.get(data[i:i + 3]) if i + 2 < n else None
        if i + 2 < n:
            table[data[i:i + 3]] = i
        if ref is not None and 0 < i - ref <= 8192:
            length = 3
            while i + length < n and length < 264 and data[ref + length] == data[i + length]:
                length += 1
            flush()
            dist, l2 = i - ref - 1, length - 2
            if l2 < 7:
                out.append((l2 << 5) | (dist >> 8))
            else:
                out.append((7 << 5) | (dist >> 8))
                out.append(l2 - 7)
            out.append(dist & 0xFF)
            i += length
        else:
            lit.append(data[i])
            i += 1
    flush()
    return bytes(out)


def write_pcd_compressed(path, pts, extra_field=True):
    """DATA binary_compressed: uint32 sizes + LZF of the struct-of-arrays payload (optionally with a 4th field)."""
    import struct
    pts = np.asarray(pts, np.float32)[:, :3]
    n = len(pts)
    cols = [pts[:, 0], pts[:, 1], pts[:, 2]]
    fields, sizes, types = "x y z", "4 4 4", "F F F"
    if extra_field:                                              # an intensity column between y and z
        cols = [pts[:, 0], pts[:, 1], np.arange(n, dtype=np.float32), pts[:, 2]]
        fields, sizes, types = "x y intensity z", "4 4 4 4", "F F F F"
    raw = b"".join(np.ascontiguousarray(c).tobytes() for c in cols)
    comp = lzf_compress(raw)
    hdr = (f"# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS {fields}\nSIZE {sizes}\nTYPE {types}\n"
           f"COUNT {' '.join('1' for _ in cols)}\nWIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\n"
           "DATA binary_compressed\n")
    with open(path, "wb") as f:
        f.write(hdr.encode())
        f.write(struct.pack("<II", len(comp), len(raw)))
        f.write(comp)
    return len(comp), len(raw)


def read_pcd_ascii(path):
    lines = open(path).read().splitlines()
    k = next(i for i, l in enumerate(lines) if l.startswith("DATA"))
    return np.array([[float(v) for v in l.split()] for l in lines[k + 1:] if l.strip()], np.float32)


def test_cli_argument_errors_exit_like_the_reference(programs):
    cli = programs[0]
    for args in ([], ["only_source.pcd"], ["-m", "notanint", "a.pcd", "b.pcd"], ["--bogus", "a.pcd", "b.pcd"], ["-r"]):
        r = subprocess.run([cli] + args, capture_output=True, text=True)
        assert r.returncode == 1 and "error:" in r.stderr and "for arg" in r.stderr
    r = subprocess.run([cli, "/nonexistent/a.pcd", "/nonexistent/b.pcd"], capture_output=True, text=True)
    assert r.returncode == 1 and "Could not load source cloud, closing" in r.stdout
    # the batch front end: no positional clouds, a readable list
    r = subprocess.run([cli, "--batch", "list.txt", "a.pcd"], capture_output=True, text=True)
    assert r.returncode == 1 and "Positional arguments are not used with --batch" in r.stderr
    r = subprocess.run([cli, "--batch", "/nonexistent/list.txt"], capture_output=True, text=True)
    assert r.returncode == 1 and "Could not read the pair list" in r.stdout
    r = subprocess.run([cli, "--batch", "list.txt", "--lanes", "0"], capture_output=True, text=True)
    assert r.returncode == 1 and "--lanes must be at least 1" in r.stderr
    # what the batch entry point cannot honour is refused, not dropped (round-2 advisor): filters, ground truth, report
    for extra, flag in ((["-s", "0.5"], "-s"), (["-t", "0.5"], "-t"), (["-g", "gt.pcd"], "-g"), (["--dump"], "--dump")):
        r = subprocess.run([cli, "--batch", "list.txt"] + extra, capture_output=True, text=True)
        assert r.returncode == 1 and "single-pair form" in r.stderr and f"for arg {flag}" in r.stderr, r.stderr


def test_cli_refuses_malformed_pcd_headers(programs, tmp_path):
    """pcl::io::loadPCDFile returns -1 on anything it cannot read and the CLI exits 1 (..._ex.cc:113-116): a
    malformed header must not abort the process or allocate what the file could never hold (no GPU needed: the
    clouds are loaded before the device is touched)."""
    good = tmp_path / "good.pcd"
    write_pcd(good, np.zeros((4, 3), np.float32))
    hdr = "VERSION 0.7\nFIELDS x y z\nSIZE {size}\nTYPE F F F\nCOUNT {count}\nWIDTH {w}\nHEIGHT 1\nPOINTS {pts}\nDATA {mode}\n"
    cases = dict(
        bad_number=hdr.format(size="4 4 4", count="1 1 1", w="abc", pts="4", mode="ascii") + "0 0 0\n" * 4,
        missing_token=hdr.format(size="4 4 4", count="1 1 1", w="4", pts="", mode="ascii") + "0 0 0\n" * 4,
        negative_size=hdr.format(size="-4 4 4", count="1 1 1", w="4", pts="4", mode="binary") + "\0" * 48,
        zero_count=hdr.format(size="4 4 4", count="0 1 1", w="4", pts="4", mode="binary") + "\0" * 48,
        huge_points=hdr.format(size="4 4 4", count="1 1 1", w="4", pts="900000000000", mode="binary") + "\0" * 48,
        huge_points_ascii=hdr.format(size="4 4 4", count="1 1 1", w="4", pts="900000000000", mode="ascii") + "0 0 0\n",
        negative_points=hdr.format(size="4 4 4", count="1 1 1", w="4", pts="-5", mode="ascii") + "0 0 0\n",
    )
    blob = hdr.format(size="4 4 4", count="1 1 1", w="4", pts="4", mode="binary_compressed").encode()
    cases_bin = dict(huge_compressed=blob + np.array([0xFFFFFFF0, 48], np.uint32).tobytes() + b"\0" * 16)
    for name, text in list(cases.items()) + list(cases_bin.items()):
        path = tmp_path / f"{name}.pcd"
        path.write_bytes(text if isinstance(text, bytes) else text.encode())
        r = subprocess.run([programs[0], str(path), str(good)], capture_output=True, text=True, cwd=tmp_path, timeout=120)
        assert r.returncode == 1 and "Could not load source cloud, closing" in r.stdout, (name, r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
def test_cpp_api_restated_reference_tests(programs):
    r = subprocess.run([programs[1]], capture_output=True, text=True, timeout=600)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failed" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("gauss", [False, True])
def test_cli_end_to_end_matches_oracle(programs, tmp_path, gauss):
    cli = programs[0]
    src, tgt, Rgt, tgt_t = synth.make_pair(3000, cfg=1, stride=3)
    write_pcd(tmp_path / "scan_a.pcd", src, binary=False)
    write_pcd(tmp_path / "scan_b.pcd", tgt, binary=True)
    gt = (src.astype(np.float64) @ Rgt.T + tgt_t).astype(np.float32)
    write_pcd(tmp_path / "gt.pcd", gt)
    args = [cli, "-r", "1.0", "-m", "5", "-i", "7", "-c", "0", "-v", "--dump", "-g", str(tmp_path / "gt.pcd")]
    if gauss:
        args.append("-u")
    args += [str(tmp_path / "scan_a.pcd"), str(tmp_path / "scan_b.pcd")]
    r = subprocess.run(args, capture_output=True, text=True, cwd=tmp_path, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    assert ("Using gaussian model" in out) == gauss
    assert "Radius of the neighborhood search: 1" in out and "Max number of neighbours: 5" in out
    assert "Transformation history:" in out and "Saving aligned source cloud to: aligned_scan_a.pcd" in out
    assert "Saving registration report to: scan_a_scan_b_summary.txt" in out
    assert re.search(r"MSE w\.r\.t\. ground truth: ", out)
    hist = re.findall(r"^T: (.*) \|\|\| R: (.*)$", out, flags=re.M)
    assert len(hist) == 7                                      # -c 0 -i 7 -> exactly 7 outer iterations
    t_cli = np.array([float(v) for v in hist[-1][0].split(",")])
    q_cli = np.array([float(v) for v in hist[-1][1].split(",")])   # x, y, z, w
    dof = float("inf") if gauss else 5.0
    ora = po.align(src, tgt, 1.0, 5, dof, 7, cost_drop_thresh=0.0, inner_max_steps=100, f_tol=10e-6)
    R_cli = po.quat_to_R([q_cli[3], q_cli[0], q_cli[1], q_cli[2]])
    # stdout carries 6 significant digits
    assert np.linalg.norm(t_cli - ora["history"][-1][:, 3]) < 5e-6
    assert synth.rotation_angle(R_cli, ora["history"][-1][:, :3]) < 5e-6
    # aligned cloud (written only because of -v) = original source moved by the final transform
    aligned = read_pcd_ascii(tmp_path / "aligned_scan_a.pcd")
    exp = src.copy()
    po.transform_cloud(exp, np.vstack([ora["history"][-1], [0, 0, 0, 1]]))
    np.testing.assert_allclose(aligned, exp, atol=5e-6)
    # summary file: 3 header lines + column header + one row per iteration
    rep = open(tmp_path / "scan_a_scan_b_summary.txt").read().splitlines()
    assert rep[0].startswith("Source: ") and rep[1].startswith("Target:") and rep[2].startswith("dof: ")
    assert rep[3].startswith("iter, n_success_steps, initial_cost, final_cost, tx, ty, tz, roll, pitch, yaw")
    assert len(rep) == 4 + 7
    row = [float(v) for v in rep[-1].split(",")]
    assert row[0] == 6 and abs(row[4] - t_cli[0]) < 1e-5
    np.testing.assert_allclose([row[2], row[3]], ora["costs"][-1], rtol=1e-5)


@pytest.mark.gpu
def test_cli_reads_binary_compressed_pcd(programs, tmp_path):
    """DATA binary_compressed (LZF, struct-of-arrays payload, extra field in the middle): same result as the same
    clouds given as plain binary files."""
    cli = programs[0]
    src, tgt, _, _ = synth.make_pair(2500, cfg=1, stride=3)
    src = np.round(src * 8) / 8                                  # repetitive bytes: the encoder emits back references
    tgt = np.round(tgt * 8) / 8
    nc, nr = write_pcd_compressed(tmp_path / "a.pcd", src, extra_field=True)
    assert nc < nr                                               # really compressed
    write_pcd_compressed(tmp_path / "b.pcd", tgt, extra_field=False)
    write_pcd(tmp_path / "a_plain.pcd", src, binary=True)
    write_pcd(tmp_path / "b_plain.pcd", tgt, binary=True)
    outs = []
    for a, b in (("a.pcd", "b.pcd"), ("a_plain.pcd", "b_plain.pcd")):
        r = subprocess.run([cli, "-r", "1", "-m", "5", "-i", "3", "-c", "0", "-v", str(tmp_path / a), str(tmp_path / b)],
                           capture_output=True, text=True, cwd=tmp_path, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(re.findall(r"^T: .*$", r.stdout, flags=re.M))
    assert len(outs[0]) == 3 and outs[0] == outs[1]
    np.testing.assert_array_equal(read_pcd_ascii(tmp_path / "aligned_a.pcd"), read_pcd_ascii(tmp_path / "aligned_a_plain.pcd"))
    # a corrupt stream is refused like any unreadable cloud
    blob = bytearray(open(tmp_path / "a.pcd", "rb").read())
    blob[-40:] = b"\xff" * 40
    open(tmp_path / "bad.pcd", "wb").write(blob)
    r = subprocess.run([cli, str(tmp_path / "bad.pcd"), str(tmp_path / "b.pcd")], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 1 and "Could not load source cloud, closing" in r.stdout


@pytest.mark.gpu
def test_cli_without_verbose_writes_nothing(programs, tmp_path):
    src, tgt, _, _ = synth.make_pair(1500, cfg=1, stride=3)
    write_pcd(tmp_path / "a.pcd", src)
    write_pcd(tmp_path / "b.pcd", tgt)
    r = subprocess.run([programs[0], "-r", "1", "-m", "5", "-i", "3", str(tmp_path / "a.pcd"), str(tmp_path / "b.pcd")],
                       capture_output=True, text=True, cwd=tmp_path, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert sorted(os.listdir(tmp_path)) == ["a.pcd", "b.pcd"]     # aligned_* only with -v, summary only with --dump


@pytest.mark.gpu
def test_cli_with_voxel_filters_ground_truth_and_report(programs, tmp_path):
    """-s / -t down-sample on the device, the association runs on the filtered source while the FULL source is
    moved along (cc:110-112) and reported on (cc:114-122): emulate the whole pipeline with the oracle."""
    cli = programs[0]
    src, tgt, Rgt, tgt_t = synth.make_pair(12000, cfg=1, stride=3)
    gt = (src.astype(np.float64) @ Rgt.T + tgt_t).astype(np.float32)
    write_pcd(tmp_path / "a.pcd", src, binary=True)
    write_pcd(tmp_path / "b.pcd", tgt, binary=True)
    write_pcd(tmp_path / "gt.pcd", gt, binary=True)
    n_it = 5
    r = subprocess.run([cli, "-r", "1.5", "-m", "6", "-i", str(n_it), "-c", "0", "-s", "0.9", "-t", "0.8", "-v", "--dump",
                        "-g", str(tmp_path / "gt.pcd"), str(tmp_path / "a.pcd"), str(tmp_path / "b.pcd")],
                       capture_output=True, text=True, cwd=tmp_path, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    # oracle emulation
    fs, ft = po.voxel_filter(src, 0.9), po.voxel_filter(tgt, 0.8)
    assert 0 < fs.shape[0] < src.shape[0] and 0 < ft.shape[0] < tgt.shape[0]
    ora = po.align(fs, ft, 1.5, 6, 5.0, n_it, cost_drop_thresh=0.0, inner_max_steps=100, f_tol=10e-6)
    full, prev = src.copy(), src.copy()
    mse_prev, mse_gt = [], []
    Tcum_prev = np.eye(4)
    for k in range(n_it):
        Tcum = np.vstack([ora["history"][k], [0, 0, 0, 1]])
        Tk = Tcum @ np.linalg.inv(Tcum_prev)                 # incremental transform of iteration k
        Tcum_prev = Tcum
        po.transform_cloud(full, Tk)
        mse_gt.append(po.calculate_mse(full, gt))
        mse_prev.append(po.calculate_mse(full, prev))
        prev = full.copy()
    rep = open(tmp_path / "a_b_summary.txt").read().splitlines()
    assert len(rep) == 4 + n_it
    for k in range(n_it):
        row = [float(v) for v in rep[4 + k].split(",")]
        np.testing.assert_allclose([row[2], row[3]], ora["costs"][k], rtol=1e-5)
        np.testing.assert_allclose(row[4:7], ora["history"][k][:, 3], atol=2e-6, rtol=1e-5)
        assert abs(row[10] - mse_prev[k]) < 2e-6 * max(1.0, mse_prev[k]) + 1e-7      # six significant digits in the file
        assert abs(row[11] - mse_gt[k]) < 2e-6 * max(1.0, mse_gt[k]) + 1e-7
    m = re.findall(r"MSE w\.r\.t\. ground truth: ([0-9.eE+-]+)", r.stdout)
    assert abs(float(m[-1]) - mse_gt[-1]) < 2e-6
    aligned = read_pcd_ascii(tmp_path / "aligned_a.pcd")     # the FULL source, moved
    assert aligned.shape[0] == src.shape[0]
    np.testing.assert_allclose(aligned, full, atol=2e-5)


@pytest.mark.gpu
def test_cli_batch_front_end_matches_single_runs(programs, tmp_path):
    """--batch list.txt: every pair of the list through ppcr_batch_run on all visible devices; each printed transform
    equals the one the ordinary single-pair command line prints for the same pair and options."""
    cli = programs[0]
    lines = ["# pairs"]
    for p in range(3):
        src, tgt, _, _ = synth.make_pair(3000 + 700 * p, cfg=5, pair=p, stride=3)
        write_pcd(tmp_path / f"s{p}.pcd", src, binary=True)
        write_pcd(tmp_path / f"t{p}.pcd", tgt, binary=bool(p % 2))
        lines.append(f"s{p}.pcd t{p}.pcd")
    lines.insert(2, "")
    (tmp_path / "pairs.txt").write_text("\n".join(lines) + "\n")
    common = ["-r", "1", "-m", "8", "-i", "4", "-c", "0"]
    r = subprocess.run([cli, "--batch", "pairs.txt", "--lanes", "2", "-v"] + common, capture_output=True, text=True,
                       cwd=tmp_path, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    got = re.findall(r"^pair (\d+) \(s\d\.pcd -> t\d\.pcd\), 4 iterations: (T: .*)$", r.stdout, flags=re.M)
    assert [int(k) for k, _ in got] == [0, 1, 2]
    for p in range(3):
        one = subprocess.run([cli, "-v"] + common + [f"s{p}.pcd", f"t{p}.pcd"], capture_output=True, text=True,
                             cwd=tmp_path, timeout=600)
        assert one.returncode == 0, one.stdout + one.stderr
        last = re.findall(r"^T: .*$", one.stdout, flags=re.M)[-1]
        a = np.array([float(v) for v in re.split(r"[,|R:T ]+", got[p][1]) if v])
        b = np.array([float(v) for v in re.split(r"[,|R:T ]+", last) if v])
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-6)      # six significant digits are printed
    # one process per GPU: the same list as rank 0 of a world of one (this rank's share through ppcr_batch_run on its
    # device, then the native RCCL all-gather of the transforms, ppcr_gather_transforms) prints the same lines
    rk = subprocess.run([cli, "--batch", "pairs.txt", "--lanes", "2", "--rank", "0", "--world", "1", "--rendezvous", "rv.id"] + common,
                        capture_output=True, text=True, cwd=tmp_path, timeout=600)
    assert rk.returncode == 0, rk.stdout + rk.stderr
    got_rk = re.findall(r"^pair (\d+) \(s\d\.pcd -> t\d\.pcd\), 4 iterations: (T: .*)$", rk.stdout, flags=re.M)
    assert got_rk == got and not (tmp_path / "rv.id").exists()
    # a record left behind by a crashed launch (any 128 bytes, or a well-formed one of another launch) is never taken for
    # this launch's: rank 0 replaces it before anybody joins; the ranks compare the --run-id token
    (tmp_path / "rv.id").write_bytes(b"\x7f" * 128)
    rk2 = subprocess.run([cli, "--batch", "pairs.txt", "--lanes", "2", "--rank", "0", "--world", "1", "--rendezvous", "rv.id",
                          "--run-id", "launch-42"] + common, capture_output=True, text=True, cwd=tmp_path, timeout=600)
    assert rk2.returncode == 0, rk2.stdout + rk2.stderr
    assert re.findall(r"^pair (\d+) \(s\d\.pcd -> t\d\.pcd\), 4 iterations: (T: .*)$", rk2.stdout, flags=re.M) == got
    # a rank that fails AFTER the communicator exists still takes part in the gathers (no rank may be left inside a
    # collective) and the launch as a whole reports the failure
    (tmp_path / "bad_ranked.txt").write_text("s0.pcd t0.pcd\ns1.pcd nowhere.pcd\n")
    rk3 = subprocess.run([cli, "--batch", "bad_ranked.txt", "--rank", "0", "--world", "1", "--rendezvous", "rv.id"] + common,
                         capture_output=True, text=True, cwd=tmp_path, timeout=120)
    assert rk3.returncode == 1 and "Could not load nowhere.pcd" in rk3.stdout and "rank 0 failed" in rk3.stderr
    assert "pair 0" not in rk3.stdout and not (tmp_path / "rv.id").exists()
    bad = subprocess.run([cli, "--batch", "pairs.txt", "--rank", "1", "--world", "2"], capture_output=True, text=True, cwd=tmp_path)
    assert bad.returncode != 0 and "--rendezvous" in bad.stderr
    # a list that names a missing cloud fails like the single-pair command
    (tmp_path / "bad.txt").write_text("s0.pcd nowhere.pcd\n")
    r = subprocess.run([cli, "--batch", "bad.txt"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 1 and "Could not load nowhere.pcd, closing" in r.stdout
