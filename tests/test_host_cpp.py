"""C++17 host front end (scpp_amd/host): the reference's SC_oneshot / SC_sim executables over the C ABI.
On this GPU-less machine the executables are linked against the CPU wave-emulation build of the kernels
(`make emu`); the product binaries link libscpp_hip.so.  Outputs are the reference's CSV tree (6 significant
digits), compared with the oracle run of the same instance."""
import glob
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "scpp_amd", "host")
CONFIG = os.path.join(ROOT, "scpp_amd", "config")


pytestmark = pytest.mark.xdist_group("host_cpp")  # one build directory (`make -C scpp_amd/host emu`): keep the module on one worker


@pytest.fixture(scope="module")
def host_emu(emu_lib):
    subprocess.check_call(["make", "-s", "-C", HOST, "emu"])
    return HOST


def _read(path):
    return np.loadtxt(path, delimiter=",", ndmin=2)


def test_sc_oneshot_writes_reference_output_tree(oracle, host_emu, tmp_path):
    K = 8
    out = subprocess.check_output([os.path.join(host_emu, "sc_oneshot_emu"), "--K", str(K), "--config", CONFIG, "--out", str(tmp_path)], text=True)
    assert "No convergence after 15 iterations." in out or "Converged after" in out
    runs = glob.glob(str(tmp_path / "output" / "RocketQuat" / "SC" / "*"))
    assert len(runs) == 1
    iters = sorted(int(os.path.basename(d)) for d in glob.glob(os.path.join(runs[0], "*")))
    sc = oracle.SC(oracle.ROCKETQUAT, K=K); sc.set_solver(1); sc.solve()
    m = sc.meta()
    assert iters == list(range(m["n_all_td"]))  # initial guess + one directory per iteration (SC_oneshot.cpp:38-62)
    # every iterate, redimensionalised like getAllSolutions (SCAlgorithm.cpp:217-232), at the CSV's 6 significant digits
    scales = sc.x_init()
    ms, rs = scales[0], np.linalg.norm(scales[1:4])
    for it in (0, 1, m["n_all_td"] - 1):
        X, U, t = sc.iterate(it)
        X = X.copy(); U = U.copy()
        X[:, 0] *= ms; X[:, 1:7] *= rs; U[:, :3] *= ms * rs; U[:, 3] *= ms * rs * rs
        Xf, Uf = _read(os.path.join(runs[0], str(it), "X.txt")), _read(os.path.join(runs[0], str(it), "U.txt"))
        tf = float(open(os.path.join(runs[0], str(it), "t.txt")).read())
        assert Xf.shape == (K, 14) and Uf.shape == (K, 4)
        assert np.allclose(Xf, X, rtol=2e-5, atol=2e-5 * np.abs(X).max())
        assert np.allclose(Uf, U, rtol=2e-5, atol=2e-5 * np.abs(U).max())
        assert abs(tf - t) <= 2e-5 * abs(t)


def test_sc_oneshot_zero_order_hold_and_fixed_time_configuration(oracle, host_emu, tmp_path):
    """A configuration with `interpolate_input false` and `free_final_time false` (and no weight_trust_region_time entry, which
    the reference only reads for a free final time, SCAlgorithm.cpp:42-45) through the C++ front end: U.txt has K - 1 rows
    (trajectoryData.hpp:27-32), t.txt keeps the configured final time, the trajectory is the oracle's literal run."""
    import re
    import shutil

    K = 8
    cfg = tmp_path / "config"
    shutil.copytree(CONFIG, cfg)
    p = cfg / "RocketQuat" / "SC.info"
    txt = re.sub(r"interpolate_input(\s+)true", r"interpolate_input\1false", p.read_text())
    txt = re.sub(r"free_final_time(\s+)true", r"free_final_time\1false", txt)
    txt = "\n".join(l for l in txt.splitlines() if not l.startswith("weight_trust_region_time"))
    p.write_text(txt + "\n")
    subprocess.check_output([os.path.join(host_emu, "sc_oneshot_emu"), "--K", str(K), "--config", str(cfg), "--out", str(tmp_path)], text=True)
    run = glob.glob(str(tmp_path / "output" / "RocketQuat" / "SC" / "*"))[0]
    last = max(int(os.path.basename(d)) for d in glob.glob(os.path.join(run, "*")))
    Xf, Uf = _read(os.path.join(run, str(last), "X.txt")), _read(os.path.join(run, str(last), "U.txt"))
    assert Xf.shape == (K, 14) and Uf.shape == (K - 1, 4)
    assert float(open(os.path.join(run, str(last), "t.txt")).read()) == 12.0
    sc = oracle.SC(oracle.ROCKETQUAT, K=K, config_root=str(cfg)); sc.set_solver(0)
    assert sc.solve() == 0
    X, U, t = sc.solution()
    assert U.shape == (K - 1, 4) and t == 12.0
    assert np.allclose(Xf, X, rtol=2e-5, atol=2e-5 * np.abs(X).max())
    assert np.allclose(Uf, U, rtol=1e-3, atol=1e-3 * np.abs(U).max())  # CSV: 6 significant digits; two independent 15-iteration runs


def test_sc_oneshot_batch_and_sc_sim(oracle, host_emu, tmp_path):
    K = 8
    out = subprocess.check_output([os.path.join(host_emu, "sc_oneshot_emu"), "--K", str(K), "--batch", "3", "--config", CONFIG, "--out", str(tmp_path)], text=True)
    assert "batch 3:" in out and "solver failures 0" in out
    out = subprocess.check_output([os.path.join(host_emu, "sc_sim_emu"), "--K", str(K), "--steps", "2", "--config", CONFIG, "--out", str(tmp_path)], text=True)
    assert "Average frequency" in out
    run = glob.glob(str(tmp_path / "output" / "RocketQuat" / "SC_sim" / "*" / "0"))[0]
    Xf, Uf = _read(os.path.join(run, "X.txt")), _read(os.path.join(run, "U.txt"))
    sc = oracle.SC(oracle.ROCKETQUAT, K=K); sc.set_solver(1)
    o = sc.sim(0.05, 2)
    assert Xf.shape == (2, 14) and Uf.shape == (2, 4)
    assert np.allclose(Xf, o["X_sim"], rtol=2e-5, atol=2e-5 * np.abs(o["X_sim"]).max())
    assert np.allclose(Uf, o["U_sim"], rtol=2e-5, atol=2e-5 * np.abs(o["U_sim"]).max())
    assert abs(float(open(os.path.join(run, "t.txt")).read()) - 0.1) < 1e-12


@pytest.mark.gpu
def test_host_executables_on_gpu(oracle, tmp_path):
    """The product binaries (linked against libscpp_hip.so) on the real device: SC_oneshot of the shipped K=50
    scenario against the oracle, a 512-instance batch and a short batched SC_sim."""
    import __graft_entry__ as g

    g.build_host()
    exe = os.path.join(HOST, "sc_oneshot")
    out = subprocess.check_output([exe, "--config", CONFIG, "--out", str(tmp_path)], text=True)
    run = glob.glob(str(tmp_path / "output" / "RocketQuat" / "SC" / "*"))[0]
    sc = oracle.SC(oracle.ROCKETQUAT, K=50); sc.set_solver(1); sc.solve()
    m = sc.meta()
    last = m["n_all_td"] - 1
    X, U, t = sc.solution()
    Xf, Uf = _read(os.path.join(run, str(last), "X.txt")), _read(os.path.join(run, str(last), "U.txt"))
    assert np.allclose(Xf, X, rtol=2e-5, atol=2e-5 * np.abs(X).max())
    assert np.allclose(Uf, U, rtol=2e-5, atol=2e-5 * np.abs(U).max())
    out = subprocess.check_output([exe, "--batch", "512", "--config", CONFIG, "--out", str(tmp_path)], text=True)
    assert "batch 512:" in out and "solver failures 0" in out
    out = subprocess.check_output([os.path.join(HOST, "sc_sim"), "--batch", "64", "--steps", "3", "--config", CONFIG, "--out", str(tmp_path)], text=True)
    assert "64 closed loops, 192 solves" in out
    # BASELINE configs[0]: the reference's default build (activeModel.hpp:10 = Rocket2d) of SC_oneshot, K = 30
    _check_rocket2d_oneshot(oracle, os.path.join(HOST, "sc_oneshot_rocket2d"), tmp_path)


def _check_rocket2d_oneshot(oracle, exe, tmp_path, K=30):
    out = subprocess.check_output([exe, "--K", str(K), "--config", CONFIG, "--out", str(tmp_path)], text=True)
    sc = oracle.SC(oracle.ROCKET2D, K=K); sc.solve()
    m = sc.meta()
    assert m["converged"] == 1 and ("Converged after %d iterations." % m["iterations"]) in out
    run = glob.glob(str(tmp_path / "output" / "Rocket2D" / "SC" / "*"))[0]
    iters = sorted(int(os.path.basename(d)) for d in glob.glob(os.path.join(run, "*")))
    assert iters == list(range(m["n_all_td"]))
    X, U, t = sc.solution()
    last = m["n_all_td"] - 1
    Xf, Uf = _read(os.path.join(run, str(last), "X.txt")), _read(os.path.join(run, str(last), "U.txt"))
    assert Xf.shape == (K, 6) and Uf.shape == (K, 2)
    assert np.allclose(Xf, X, rtol=2e-5, atol=2e-5 * np.abs(X).max())
    assert np.allclose(Uf, U, rtol=1e-4, atol=1e-4 * np.abs(U).max())
    assert abs(float(open(os.path.join(run, str(last), "t.txt")).read()) - t) <= 2e-5 * t


def test_sc_oneshot_rocket2d_default_model(oracle, host_emu, tmp_path):
    """The front end compiled for the reference's DEFAULT active model (-DSCPP_ACTIVE_MODEL_ROCKET2D): Rocket2D SC_oneshot through
    the device solver instantiated for Rocket2d's constraint table, output tree under output/Rocket2D/SC, against the oracle's
    literal reference-shaped run (CPU: emulation build of the kernels, K = 12)."""
    _check_rocket2d_oneshot(oracle, os.path.join(host_emu, "sc_oneshot_rocket2d_emu"), tmp_path, K=12)


def test_sc_oneshot_gpus_shards_equal_single_device(host_emu, tmp_path):
    """`sc_oneshot --batch B --gpus N`: one host thread + one device context per GPU, contiguous static shards (uneven here),
    results concatenated in instance order -- the same trajectories as the single-context run (checksum over every state)."""
    exe = os.path.join(host_emu, "sc_oneshot_emu")
    one = subprocess.check_output([exe, "--K", "8", "--batch", "5", "--config", CONFIG, "--out", str(tmp_path)], text=True)
    two = subprocess.check_output([exe, "--K", "8", "--batch", "5", "--gpus", "2", "--config", CONFIG, "--out", str(tmp_path)], text=True)
    cs = lambda t: [l for l in t.splitlines() if l.startswith("checksum X")][0]
    assert "batch 5 on 2 GPUs:" in two and "solver failures 0" in two
    assert cs(one) == cs(two)


def test_sc_oneshot_scvx_mode(oracle, host_emu, tmp_path):
    """`sc_oneshot --scvx`: the C++ SCvxAlgorithm front end over scpp_hip_scvx_*, against the oracle's SCvx run."""
    K = 10
    out = subprocess.check_output([os.path.join(host_emu, "sc_oneshot_emu"), "--scvx", "--K", str(K), "--config", CONFIG, "--out", str(tmp_path)], text=True)
    s = oracle.SCvx(K=K); s.set_solver(1)
    assert s.solve() == 0
    m = s.meta()
    assert f"converged {m['converged']}, solver failures 0, mean iterations {m['iterations']:.2f}, mean sub-problem solves {m['solves']:.2f}" in out
    # round 6: every trajectory of getAllSolutions is written, like the reference's driver does (SC_oneshot.cpp:31-63): <time>/0 = the initial
    # trajectory, <time>/j = the trajectory after iteration j
    base = glob.glob(str(tmp_path / "output" / "RocketQuat" / "SCvx" / "*"))[0]
    dirs = sorted(int(d) for d in os.listdir(base))
    assert dirs == list(range(m["iterations"] + 1)) and f"({m['iterations'] + 1} iterates)" in out
    # the oracle keeps all_td nondimensional (as the reference's member does); getAllSolutions redimensionalises (SCvxAlgorithm.cpp:247-255):
    # mass and length scale from the final trajectory, which the oracle returns in both forms
    (Xd, _, _), (Xn, _, _) = s.iterate(-1), s.iterate(dirs[-1])
    ms, rs = Xd[0, 0] / Xn[0, 0], Xd[0, 3] / Xn[0, 3]
    fx = np.array([ms] + [rs] * 6 + [1.0] * 7)
    fu = np.array([ms * rs] * 3 + [ms * rs * rs])
    for j in dirs:
        run = os.path.join(base, str(j))
        X, U, t = s.iterate(j)
        X, U = X * fx, U * fu
        Xf, Uf = _read(os.path.join(run, "X.txt")), _read(os.path.join(run, "U.txt"))
        assert np.allclose(Xf, X, rtol=2e-5, atol=2e-5 * np.abs(X).max()), j
        assert np.allclose(Uf, U, rtol=2e-4, atol=2e-4 * np.abs(U).max()), j
        assert abs(float(open(os.path.join(run, "t.txt")).read()) - t) <= 1e-5 * t
    assert np.allclose(Xd, Xn * fx, rtol=1e-12)


def test_mpc_sim_matches_oracle_closed_loop(oracle, host_emu, tmp_path):
    """scpp_amd/host/mpc_sim (MPC_sim.cpp:16-129): the host-stepped single loop against oracle/mpc.hpp, the reference's
    reduced output tree, and the all-device batch mode."""
    steps = 40
    out = subprocess.check_output([os.path.join(host_emu, "mpc_sim_emu"), "--steps", str(steps), "--config", CONFIG, "--out", str(tmp_path)], text=True)
    assert "Failed solves: 0" in out
    final = np.array([float(v) for v in out.split("Final state:")[1].split("\n")[0].split()])
    o = oracle.MPC()
    q = o.sim(o.x_init, max_steps=steps)
    assert q["steps"] == steps and np.abs(final - q["x"]).max() <= 2e-6 * np.abs(q["x"]).max()
    run = glob.glob(str(tmp_path / "output" / "Rocket2D" / "MPC" / "*" / "0"))[0]
    Xf, Uf, tf = _read(os.path.join(run, "X.txt")), _read(os.path.join(run, "U.txt")), _read(os.path.join(run, "t.txt"))
    assert Xf.shape == (30, 6) and Uf.shape == (30, 2) and tf.shape == (30, 1)     # write_steps = 30 (MPC_sim.cpp:24)
    assert np.allclose(tf[:, 0], 0.01 * (1 + np.arange(30)), rtol=1e-5)            # stride 40 / 30 = 1
    assert np.allclose(Xf[0], o.sim(o.x_init, max_steps=1)["x"], rtol=2e-5, atol=1e-4)
    out = subprocess.check_output([os.path.join(host_emu, "mpc_sim_emu"), "--batch", "3", "--steps", "5", "--config", CONFIG], text=True)
    assert "3 closed loops, 15 controller steps, 0 failed solves" in out
    # the shipped model.info (constrain_initial_final true) is refused, as its own comment demands for MPC
    r = subprocess.run([os.path.join(host_emu, "mpc_sim_emu"), "--keep-constraint", "--config", CONFIG], capture_output=True, text=True)
    assert r.returncode == 1 and "constrain_initial_final" in r.stderr


@pytest.mark.gpu
def test_cpp_host_rccl_all_gather_on_gpu(model, hip_lib):
    """The torch-free multi-GPU driver (host/scvx_multi_gpu.cpp: one process, one context + one RCCL communicator per device,
    ncclAllGather of the device-resident result rows in chunks) on the one GPU of the test box: 96 SCvx instances through 32
    slots, gathered with 3 collectives; the executable itself checks gathered == shard rows bitwise (exit code), and its
    converged count / checksum must equal the Python front end's streaming run of the same instances."""
    import re

    import __graft_entry__ as g
    import scpp_amd

    g.build_host()
    exe = os.path.join(HOST, "scvx_multi_gpu")
    N = 96
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe, "--batch", str(N), "--gpus", "1", "--slots", "32", "--chunk-mb", "0.25", "--config", CONFIG], capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bitwise: yes" in r.stdout and "3 collective(s)" in r.stdout, r.stdout
    conv = int(re.search(r"converged (\d+) \(engine counters (\d+)\)", r.stdout).group(1))
    cs = float(re.search(r"checksum X (\S+)", r.stdout).group(1))
    alg = scpp_amd.SCvxAlgorithm(model, K=50, batch_max=32, library=hip_lib).initialize()
    n = alg.solveStream(model.randomized_initial_states(N), slots=32)
    o = alg.getStreamSolution()
    assert n == conv and abs(float(o["X"].sum()) - cs) <= 1e-9 * abs(cs)
    alg.ctx.close()
    print(r.stdout.strip().replace("\n", " | "))


@pytest.mark.parametrize("batch,gpus,rowd,chunk_mb", [(65536, 8, 910, 64.0), (37, 5, 910, 0.02), (100, 4, 910, 0.06), (7, 8, 26, 0.0002),
                                                        (1000, 3, 910, 0.75)])
def test_all_gather_layout_and_reindexing_with_several_gpus(batch, gpus, rowd, chunk_mb):
    """host/scvx_multi_gpu's chunked ncclAllGather with gpus > 1, uneven shards (incl. an empty one) and >= 3 chunks, on CPU: the
    layout / re-indexing arithmetic (host/gather_layout.hpp) under the collective's semantics, every device's copy checked
    (the test boxes have one GPU, so N > 1 never ran on hardware)."""
    import subprocess

    import __graft_entry__ as g

    host = os.path.join(ROOT, "scpp_amd", "host")
    # under the build lock: another worker's `make all` (build_host) may be re-linking this very executable (seen once as "Text file busy")
    with g._BuildLock():
        subprocess.check_call(["make", "-s", "-C", host, "gather_layout_test"])
        r = subprocess.run([os.path.join(host, "gather_layout_test"), str(batch), str(gpus), str(rowd), str(chunk_mb)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "wrong entries 0" in r.stdout
    if (batch, gpus) == (37, 5):
        assert "4 chunk(s)" in r.stdout and "shards 8 8 7 7 7" in r.stdout
