"""GPU: a torch-free C++ host on libvoltrix_hip.so (harness/capi_host_example.cpp) -- hipMalloc buffers, the two-phase
builders (handle, unit table, panel plan, panel order, stage records), every form of the SpMM, the transpose and the
Cuthill-McKee search -- built with hipcc and run as a child
process; it checks its results against a CPU loop and reports through its exit code."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_binary(tmp_path_factory):
    from voltrix import capi

    capi.lib()   # builds libvoltrix_hip.so when it is missing
    out = str(tmp_path_factory.mktemp("capi_host") / "capi_host_example")
    lib_dir = os.path.join(REPO, "voltrix-spmm_amd", "lib")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(REPO, "include"),
           os.path.join(REPO, "harness", "capi_host_example.cpp"), "-L", lib_dir, "-lvoltrix_hip", f"-Wl,-rpath,{lib_dir}",
           "-o", out]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr[-3000:]
    return out


@pytest.mark.parametrize("num_nodes,mean_degree,num_feats", [(6000, 300, 128), (1003, 40, 64), (20000, 120, 256), (50000, 200, 128)])
def test_cpp_host_runs_both_formats_through_the_c_abi(cuda_device, host_binary, num_nodes, mean_degree, num_feats):
    proc = subprocess.run([host_binary, str(num_nodes), str(mean_degree), str(num_feats)], capture_output=True, text=True,
                          timeout=300)
    assert proc.returncode == 0, (proc.returncode, proc.stdout[-2000:], proc.stderr[-2000:])
    assert "rel err" in proc.stdout and "order equals the CPU search" in proc.stdout
    assert "panels in" in proc.stdout and "pieces" in proc.stdout      # step 2b: voltrix_launch_spmm_panel_parts_f16 + combine
