"""mapdn_amd/build.py: the library on disk is tied to the sources by a CONTENT hash embedded in it, and N ranks
that find it missing / stale at the same time compile it exactly once."""
import os
import stat
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_carries_the_hash_of_the_sources_on_disk():
    from mapdn_amd import _lib, build
    assert not build.stale()
    h = build.source_hash()
    assert build.library_hash() == h and len(h) == 64
    assert _lib.load().mapdn_build_info().decode() == "MAPDN_SRC_HASH=" + h


def test_an_edit_makes_the_library_stale_and_touch_does_not(tmp_path, monkeypatch):
    import shutil
    from mapdn_amd import build
    csrc = tmp_path / "mapdn_amd" / "csrc"
    shutil.copytree(build.CSRC, csrc)
    os.makedirs(tmp_path / "include"); shutil.copy(os.path.join(ROOT, "include", "mapdn.h"), tmp_path / "include" / "mapdn.h")
    monkeypatch.setattr(build, "CSRC", str(csrc))
    h0 = build.source_hash()
    assert h0 == build.library_hash()                        # a copy of the tree hashes like the tree
    os.utime(csrc / "nr_tree.hpp", (1, 1))                   # mtimes do not matter (round 4's rule was mtime-based)
    assert build.source_hash() == h0
    with open(csrc / "nr_common.hpp", "a") as f:
        f.write("// edited\n")
    assert build.source_hash() != h0 and build.stale()
    monkeypatch.setenv("MAPDN_EXTRA_FLAGS", "-DMAPDN_NR_STAMPS")
    assert build.source_hash() != h0                          # other flags are another library too
    assert build.library_hash(str(tmp_path / "nothing.so")) is None


def test_eight_ranks_build_once(tmp_path):
    """eight processes call build_locked() on a missing library at once, with a stand-in compiler that logs its invocations: one
    process compiles (4 NR parts + 6 sources + 1 link = 11 invocations), seven wait and return the same file"""
    fake = tmp_path / "hipcc"
    log = tmp_path / "calls.log"
    fake.write_text(textwrap.dedent(f"""\
        #!{sys.executable}
        import sys, time
        a = sys.argv[1:]
        out = a[a.index("-o") + 1]
        open({str(log)!r}, "a").write(("link" if "-shared" in a else "cc") + "\\n")
        time.sleep(0.2)
        if "-shared" in a:
            data = b"".join(open(x, "rb").read() for x in a if x.endswith(".o"))
        else:
            data = " ".join(a).replace('"', "").replace("-DMAPDN_SRC_HASH=", "MAPDN_SRC_HASH=").encode()
        open(out, "wb").write(data)
        """))
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    lib = tmp_path / "libfake.so"
    env = dict(os.environ, HIPCC=str(fake), MAPDN_BUILD_OUT=str(lib), PYTHONPATH=ROOT)
    code = "from mapdn_amd import build; import sys; p = build.build_locked(); assert not build.stale(); print(p)"
    procs = [subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(8)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-500:] for o in outs]
    assert all(o[0].strip() == str(lib) for o in outs)
    calls = log.read_text().split()
    assert calls.count("link") == 1 and calls.count("cc") == 13, calls          # 4 NR parts + 9 sources (round 6: critic.hip, rollout.hip, policy_bwd.hip)
    assert not os.path.exists(str(lib) + ".lock")


def test_binary_only_install_loads_the_library_it_has(monkeypatch):
    """a prebuilt library without mapdn_amd/csrc beside it (nothing to hash) must load — with a warning — instead of
    raising a raw FileNotFoundError out of the staleness check."""
    import warnings
    from mapdn_amd import _lib, build

    def gone():
        raise FileNotFoundError("csrc/plan.cpp")
    monkeypatch.setattr(build, "source_hash", gone)
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.delenv("MAPDN_LIB_PATH", raising=False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        lib = _lib.load()
    assert hasattr(lib, "mapdn_create") and any("source-hash check" in str(x.message) for x in w)


def test_hv_init_reaches_the_converter_through_the_scenario_loader(tmp_path, monkeypatch):
    """load_scenario / the env constructor pass hv_init (or MAPDN_HV_INIT) on to from_pandapower"""
    from mapdn_amd import data
    seen = []
    monkeypatch.setattr(data, "read_pandapower_pickle", lambda p: "net")
    monkeypatch.setattr(data, "from_pandapower", lambda net, hv_init="refuse": seen.append(hv_init) or "spec")
    monkeypatch.setattr(data, "load_profiles_csv", lambda *a: "prof")
    (tmp_path / "model.p").write_bytes(b"")
    assert data.load_scenario(str(tmp_path)) == ("spec", "prof")
    data.load_scenario(str(tmp_path), hv_init="flat")
    monkeypatch.setenv("MAPDN_HV_INIT", "flat")
    data.load_scenario(str(tmp_path))
    assert seen == ["refuse", "flat", "flat"]
