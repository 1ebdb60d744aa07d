"""CPU: the C-ABI library loads and exports every symbol include/msm.h declares.
No compute call is made here (there is no GPU in this tier)."""
import ctypes
import os
import re

import msm_amd
from msm_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "msm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(msm_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported():
    names = _declared()
    assert len(names) >= 20
    lib = msm_amd.load()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/msm.h but not exported"
    assert sorted(_lib.SYMBOLS) == names, "msm_amd/_lib.py binds a different set than the header declares"


def test_struct_layout_matches_header():
    # sizes the C compiler gives for the two structs of include/msm.h
    assert ctypes.sizeof(_lib.MsmConfig) == 160
    assert ctypes.sizeof(_lib.MsmStreamStatus) == 64


def test_create_rejects_bad_config_without_gpu():
    lib = msm_amd.load()
    ctx = ctypes.c_void_p()
    cfg = _lib.MsmConfig()
    cfg.struct_size = 1
    assert lib.msm_create(ctypes.byref(cfg), ctypes.byref(ctx)) == -1
    assert b"struct_size" in lib.msm_last_error(None)
    assert lib.msm_create(None, ctypes.byref(ctx)) == -1
    assert lib.msm_step(None) == -1


def test_pmass_hbar_resolution():
    # common/src/parameters.rs:222-259
    pm, hb = msm_amd.determine_pmass_hbar_(3e16, ntot=1e10, hbar_=0.01)
    assert pm == 3e16 / 1e10 and hb == 0.01
    pm, hb = msm_amd.determine_pmass_hbar_(1e11, hbar_=0.05)
    assert hb == 0.05 and pm == 1.757e-90 / 0.05
    pm, hb = msm_amd.determine_pmass_hbar_(1.0, particle_mass=2.0)
    assert pm == 2.0 and hb == 1.757e-90 / 2.0


def test_rust_binding_lists_the_same_symbols():
    # bindings/msm_hip.rs (source only: no Rust toolchain in the build image) declares exactly the header's entry points
    text = open(os.path.join(ROOT, "bindings", "msm_hip.rs")).read()
    block = text[text.index('extern "C" {'):text.index("\n}\n", text.index('extern "C" {'))]
    rust = sorted(set(re.findall(r"pub fn (msm_[a-z_0-9]+)\s*\(", block)))
    assert rust == _declared()
    # the two repr(C) structs carry the header's fields in order
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "msm.h")).read(), flags=re.S)
    for c_name, rs_name in (("msm_config", "MsmConfig"), ("msm_stream_status", "MsmStreamStatus")):
        c_body = hdr[hdr.index("typedef struct " + c_name):hdr.index("} " + c_name)]
        c_fields = re.findall(r"\b([a-z_0-9]+)\s*;", c_body)
        rs_body = text[text.index("pub struct " + rs_name):]
        rs_body = rs_body[:rs_body.index("}")]
        rs_fields = re.findall(r"pub ([a-z_0-9]+):", rs_body)
        assert rs_fields == c_fields, (c_name, rs_fields, c_fields)


def test_missing_rccl_is_an_error_code_not_a_crash():
    # the communicator calls bind librccl on first use; when no copy can be opened they must hand back
    # MSM_ERR_INVALID and a message.  The binding is cached per process, hence the child process.
    import subprocess
    import sys
    code = (
        "import ctypes, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import msm_amd\n"
        "lib = msm_amd.load()\n"
        "buf = ctypes.create_string_buffer(128)\n"
        "for _ in range(2):\n"                       # the second call reads the cached failure
        "    rc = lib.msm_comm_unique_id(buf)\n"
        "    msg = lib.msm_last_error(None).decode()\n"
        "    assert rc == -1 and msg.startswith('librccl not found: ') and len(msg) > 25, (rc, msg)\n"
        "print('clean:', msg)\n")
    env = dict(os.environ, MSM_RCCL_LIB="/nonexistent/librccl.so.1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "clean: librccl not found: /nonexistent/librccl.so.1" in r.stdout
