"""CPU: the JIT layer (voltrix/jit, voltrix/jit_kernels/tuner.py) against golden output of the REFERENCE's own
pure-Python modules (tests/golden/ref_python_goldens.json, produced by tests/golden/make_goldens.py), plus a real
hipcc build of a host-only kernel driven through Runtime like the reference's tests/test_jit.py:37-62."""
import json
import os
import re

import pytest
import torch

from conftest import GOLDEN

import voltrix
from voltrix import jit
from voltrix.jit import template as vtemplate
from voltrix.jit import compiler as vcompiler

GOLD = json.load(open(os.path.join(GOLDEN, "ref_python_goldens.json")))
TYPES = {v: k for k, v in vtemplate.typename_map.items()}

# CUDA -> HIP spellings the generated source may differ by (SURVEY.md section 8c "generated-source shape")
_DEVICE_TYPES = {"cudaStream_t": "hipStream_t", "__nv_bfloat16*": "__hip_bfloat16*", "__nv_fp8_e4m3*": "__hip_fp8_e4m3*"}


def _arg_defs(gold_defs):
    return tuple((n, TYPES[t]) for n, t in gold_defs)


def _signature_and_casts(code):
    """Everything from `extern "C"` to the end of the cast block -- the part that defines the C-ABI."""
    start = code.index('extern "C" void launch(')
    lines = code[start:].split("\n")
    keep = [lines[0], lines[1]]
    for line in lines[2:]:
        if line.startswith("    auto ") and "reinterpret_cast" in line:
            keep.append(line)
        else:
            break
    return "\n".join(keep)


@pytest.mark.parametrize("key", ["generate_test_jit", "generate_spmm", "generate_preprocess", "generate_hmat_gen",
                                 "generate_swizzle"])
def test_generate_reproduces_reference_abi(key):
    g = GOLD[key]
    arg_defs = _arg_defs(g["arg_defs"])
    body = g["body"]
    code = jit.generate((), arg_defs, body)
    ref_abi = _signature_and_casts(g["code"])
    for cuda_name, hip_name in _DEVICE_TYPES.items():
        ref_abi = ref_abi.replace(cuda_name, hip_name)
    assert _signature_and_casts(code) == ref_abi
    # body is spliced with the same 4-space indentation and the same terminator
    ref_tail = g["code"][g["code"].index("    // Cast raw types"):]
    my_tail = code[code.index("    // Cast raw types"):]
    for cuda_name, hip_name in _DEVICE_TYPES.items():
        ref_tail = ref_tail.replace(cuda_name, hip_name)
    assert my_tail == ref_tail
    assert code.startswith("// Voltrix-SpMM auto-generated JIT HIP source file\n\n#include ")
    head = code[:code.index('extern "C"')]
    assert "#include <hip/hip_runtime.h>" in head and "cuda" not in head.lower()


def test_generate_include_blocks_are_sorted_and_split():
    code = jit.generate(('"voltrix/spmm_kernels.hpp"', "<vector>", '"voltrix/bmat_kernels.hpp"'), (("n", int),), "\n")
    sys_block, pkg_block = code.split("\n\n")[1:3]
    assert sys_block.split("\n") == sorted(sys_block.split("\n")) and "#include <vector>" in sys_block
    assert pkg_block.split("\n") == ['#include "voltrix/bmat_kernels.hpp"', '#include "voltrix/spmm_kernels.hpp"']


def test_cpp_format_hash_and_constants_match_reference():
    for case in GOLD["cpp_format"][:1]:
        assert jit.cpp_format(case["template"], case["keys"]) == case["result"]
    for s, h in GOLD["hash_to_hex"].items():
        assert jit.hash_to_hex(s) == h
    for name, value in GOLD["const"].items():
        assert getattr(voltrix, name) == value  # flag names are part of the drop-in surface
    for module, names in GOLD["public_names"].items():
        mod = {"voltrix.jit": voltrix.jit, "voltrix.jit_kernels": voltrix, "voltrix.spmm": voltrix}[module]
        for n in names:
            assert hasattr(mod, n), f"{module}.{n} missing"
    assert voltrix.BLK_H == 16 and voltrix.BLK_W == 8


def test_type_maps_cover_reference_types():
    for tname, (sig_t, body_t) in GOLD["genc_map"].items():
        mine = vtemplate.genc_map[TYPES[tname]]
        assert mine[0] == sig_t
        assert mine[1] == _DEVICE_TYPES.get(body_t, body_t)
    for tname in GOLD["typename_map"].values():
        assert tname in TYPES


def test_kernel_args_round_trip_without_eval():
    arg_defs = _arg_defs(GOLD["generate_spmm"]["arg_defs"])
    text = vtemplate.args_to_text(arg_defs)
    assert text.startswith("('blk_offsets', torch.int), ('hspa_packed', torch.uint32)")
    assert tuple(vtemplate.args_from_text(text)) == arg_defs
    assert vtemplate.args_from_text("") == []


class _Capture:
    def __enter__(self):
        self.r, self.w = os.pipe()
        self.saved = os.dup(1)
        os.dup2(self.w, 1)
        return self

    def __exit__(self, *exc):
        os.dup2(self.saved, 1)
        os.close(self.w)
        with os.fdopen(self.r, "r") as f:
            self.text = f.read()


def test_build_and_call_host_kernel(tmp_path, monkeypatch):
    """The reference's JIT smoke (tests/test_jit.py): build, call launch, check what it printed and the return code.
    Tensors are CPU tensors here -- launch only prints the pointers, so no GPU is needed."""
    monkeypatch.setenv("VOLTRIX_CACHE_DIR", str(tmp_path))
    args = (("lhs", torch.float16), ("scale", torch.float), ("count", int), ("flag", bool))
    body = "\n"
    body += "std::cout << reinterpret_cast<uint64_t>(lhs) << std::endl;\n"
    body += "std::cout << reinterpret_cast<uint64_t>(scale) << std::endl;\n"
    body += "std::cout << count << std::endl;\n"
    body += "std::cout << flag << std::endl;\n"
    body += "__return_code = count == 7 ? 0 : 3;\n"
    code = jit.generate((), args, body)
    func = jit.build("test_func", args, code)
    assert os.path.dirname(func.path) == os.path.join(str(tmp_path), "cache")
    assert re.fullmatch(r"kernel\.test_func\.[0-9a-f]{12}", os.path.basename(func.path))
    assert sorted(os.listdir(func.path)) == ["kernel.args", "kernel.hip", "kernel.so"]
    h = torch.empty(1, dtype=torch.float16)
    s = torch.empty(1, dtype=torch.float32)
    with _Capture() as cap:
        rc = func(h, s, 7, True)
    assert rc == 0
    assert cap.text == f"{h.data_ptr()}\n{s.data_ptr()}\n7\n1\n"
    with _Capture():
        assert func(h, s, 8, False) == 3  # __return_code is really plumbed
    with pytest.raises(AssertionError, match="Expected 4 arguments"):
        func(h, s, 7)
    with pytest.raises(AssertionError, match="Expected tensor dtype"):
        func(s, s, 7, True)
    # second build is a cache hit on the same directory (no recompilation)
    mtime = os.path.getmtime(os.path.join(func.path, "kernel.so"))
    vcompiler.runtime_cache.cache.clear()
    again = jit.build("test_func", args, code)
    assert again.path == func.path and os.path.getmtime(os.path.join(func.path, "kernel.so")) == mtime


def test_build_failure_raises(tmp_path, monkeypatch):
    monkeypatch.setenv("VOLTRIX_CACHE_DIR", str(tmp_path))
    code = jit.generate((), (("n", int),), "\nthis is not C++;\n")
    with pytest.raises(RuntimeError, match="Failed to compile"):
        jit.build("broken", (("n", int),), code)


def test_tuner_picks_fastest_valid_and_memoises(tmp_path, monkeypatch):
    monkeypatch.setenv("VOLTRIX_CACHE_DIR", str(tmp_path))
    from voltrix.jit_kernels.tuner import JITTuner

    template = "\n__return_code = (n == {BAD}) ? 2 : 0;\n"
    arg_defs = (("n", int),)
    times = {0: 3.0, 1: 1.0, 5: 0.5}  # variant BAD=5 would be fastest but is illegal for n=5
    tuner = JITTuner()
    space = ({"BAD": 0}, {"BAD": 1}, {"BAD": 5})
    seen = []

    def bench_fn(fn):
        rc_probe = fn()
        seen.append(rc_probe)
        return times[space[(len(seen) - 1) % 2]["BAD"]]  # only the two legal variants are ever timed, in order

    rt = tuner.compile_and_tune(name="toy", keys={"k": 1}, space=space, includes=(), arg_defs=arg_defs,
                                template=template, args=(5,), bench=bench_fn)
    assert rt(5) == 0 and tuner.tuned_keys[("toy", "{'k': 1}")] == {"BAD": 1}
    assert len(seen) == 2  # the illegal variant was never timed
    assert tuner.compile_and_tune(name="toy", keys={"k": 1}, space=space, includes=(), arg_defs=arg_defs,
                                  template=template, args=(5,), bench=bench_fn) is rt
    # a fresh process (new tuner) reuses the persisted choice without timing anything
    tuner2 = JITTuner()
    rt2 = tuner2.compile_and_tune(name="toy", keys={"k": 1}, space=space, includes=(), arg_defs=arg_defs,
                                  template=template, args=(5,), bench=lambda fn: pytest.fail("re-tuned"))
    assert rt2.path == rt.path
    # different keys -> different signature -> tuned again (quirk 9: F / dtype / device are part of the key)
    tuner2.compile_and_tune(name="toy", keys={"k": 2}, space=space, includes=(), arg_defs=arg_defs,
                            template=template, args=(5,), bench=bench_fn)
    assert len(seen) == 4


def test_cache_version_follows_the_include_graph_of_the_jit_kernels():
    """get_repo_version (reference compiler.py:45-59 hashes its whole include tree): here the hash covers the headers a JIT
    kernel can include -- the entries the kernel modules name and what those include -- so editing a header of the
    ahead-of-time library alone (plan builders, unit tables, the search) does not throw the prebuilt kernel cache away."""
    from voltrix.jit import compiler

    closure = compiler.jit_header_closure()
    assert {"voltrix/spmm_kernels.hpp", "voltrix/bmat_kernels.hpp", "voltrix/traits.hpp"} <= set(closure)
    assert "voltrix/reorder_kernels.hpp" not in closure and "voltrix/unit_table.hpp" not in closure
    assert len(compiler.get_repo_version()) == 12


def test_compiler_version_is_read_without_starting_a_program(monkeypatch):
    """Round 6: loading a cached kernel must not spawn anything -- `hipcc --version` (which runs hipconfig)
    was started from inside the first spmm call, i.e. after the GPU was initialised under `rocprofv3 --pmc`.  The version comes
    from hip_version.h beside the compiler and equals what `hipcc --version` prints (the cache keys do not move)."""
    import subprocess

    path, version = vcompiler.get_hipcc_compiler()
    printed = re.search(r"HIP version:\s*([\d.]+)", subprocess.run([path, "--version"], capture_output=True, text=True).stdout).group(1)
    assert version == printed

    def no_spawn(*a, **k):
        raise AssertionError(f"spawned {a}")

    monkeypatch.setattr(subprocess, "run", no_spawn)
    monkeypatch.setattr(subprocess, "Popen", no_spawn)
    vcompiler.get_hipcc_compiler.cache_clear()
    try:
        assert vcompiler.get_hipcc_compiler() == (path, version)
        assert vcompiler.kernel_dir("probe", "int x;")
    finally:
        vcompiler.get_hipcc_compiler.cache_clear()


def test_finalists_of_a_sampled_sweep_cover_other_schedules():
    """Round 6: a sweep timed on a sample lets its best candidate and the best of every other SCHEDULE run the whole handle before
    the choice is made (the sample ranks shapes and depths reliably, schedules not: products-like relabelled, 3.33 vs 4.14 ms)."""
    from voltrix.jit_kernels.tuner import pick_finalists

    timed = [(1.00, {"FS": 64, "DEPTH": 4, "SCHED": 4}), (1.01, {"FS": 64, "DEPTH": 3, "SCHED": 4}),
             (1.05, {"FS": 64, "DEPTH": 3, "SCHED": 2}), (1.20, {"FS": 64, "DEPTH": 3, "SCHED": 3}),
             (0.99, {"FS": 128, "DEPTH": 3, "SCHED": 4})]
    got = pick_finalists(timed, 3)
    assert got[0] == {"FS": 128, "DEPTH": 3, "SCHED": 4}
    assert [g["SCHED"] for g in got] == [4, 2, 3]
    assert pick_finalists(timed[:2], 3) == [timed[0][1], timed[1][1]]          # one schedule only: the runner-up
    assert pick_finalists([(1.0, {"A": 1})], 3) == [{"A": 1}]
