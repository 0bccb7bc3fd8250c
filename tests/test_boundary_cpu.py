"""Boundary proof on a box without a GPU: the documented reference-side binding and the C++ adapter
compile against the REFERENCE'S OWN TYPES - common/alias.h (Eigen 3.3.9 and Sophus as vendored under
/root/reference/thirdparty), built with the reference's -DEIGEN_INITIALIZE_MATRICES_BY_ZERO - and the binding links and
runs against a stub of the C ABI. Needs the reference tree (authoring container); skipped where it is absent."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "thirdparty", "eigen")) or shutil.which("g++") is None,
                                reason="needs /root/reference (Eigen, Sophus, common/alias.h) and g++")
INC = ["-I" + os.path.join(REF, "common"), "-I" + os.path.join(REF, "thirdparty", "eigen"),
       "-I" + os.path.join(REF, "thirdparty", "sophus"), "-I" + os.path.join(ROOT, "include"),
       "-I" + os.path.join(ROOT, "oracle", "ref", "shim")]
DEFS = ["-DEIGEN_INITIALIZE_MATRICES_BY_ZERO", "-DSOPHUS_USE_BASIC_LOGGING", "-DGOOGLE_STRIP_LOG=1"]


def _integration_body():
    """the code block under "`Estimator::UpdateJosephForm` (src/estimator.cpp:1257-1288)" in INTEGRATION.md, verbatim"""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```cpp\n(void Estimator::UpdateJosephForm\(\) \{.*?\n\})\n```", text, re.S)
    assert m, "INTEGRATION.md no longer carries the UpdateJosephForm binding"
    return m.group(1)


def test_documented_binding_compiles_links_and_runs_against_the_reference_types(tmp_path):
    body = _integration_body()
    assert "xivo_hip_update_joseph_host" in body and "#ifdef USE_HIP_UPDATE" in body
    (tmp_path / "integration_body.inc").write_text(body + "\n")
    exe = str(tmp_path / "integration")
    src = os.path.join(ROOT, "tests", "boundary")
    subprocess.run(["gcc", "-c", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(src, "stub_xivo_hip.c"), "-o",
                    str(tmp_path / "stub.o")], check=True)
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-w"] + DEFS + INC + ["-I" + str(tmp_path), os.path.join(src, "integration_main.cpp"),
                        str(tmp_path / "stub.o"), "-o", exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    run = subprocess.run([exe], stdout=subprocess.PIPE, text=True)
    assert run.returncode == 0 and "calls=64 ok=1" in run.stdout, run.stdout


CALIB_DEFS = ["-DUSE_ONLINE_TEMPORAL_CALIB", "-DUSE_ONLINE_IMU_CALIB", "-DUSE_ONLINE_CAMERA_CALIB"]


@pytest.mark.parametrize("calib", [[], CALIB_DEFS, CALIB_DEFS[:1], CALIB_DEFS[2:]])
def test_adapter_compiles_with_the_reference_matrix_types(tmp_path, calib):
    """xivo_amd/host/estimator_hip.cpp with -DXIVO_HIP_USE_EIGEN: MatX / VecX / Vec2 / Vec3 / Mat3 are common/alias.h's - in the
    default build and with the reference's online-calibration defines (src/CMakeLists.txt:13-15: all three, temporal only,
    camera only), whose GPU run is tests/test_host_adapter_gpu.py::test_online_calibration_build_of_the_cpp_adapter."""
    obj = str(tmp_path / "estimator_hip_eigen.o")
    r = subprocess.run(["g++", "-std=c++17", "-O0", "-w", "-c", "-DXIVO_HIP_USE_EIGEN"] + calib + DEFS + INC +
                       [os.path.join(ROOT, "xivo_amd", "host", "estimator_hip.cpp"), "-o", obj],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    syms = subprocess.run(["nm", "-C", obj], stdout=subprocess.PIPE, text=True).stdout
    for name in ("xivo::hip::Estimator::UpdateJosephForm()", "xivo::hip::Estimator::MHGating()", "xivo::hip::Estimator::FilterUpdate()",
                 "xivo::hip::Estimator::ComputeInstateJacobians()", "xivo::hip::Estimator::Propagate(bool, double)",
                 "xivo::hip::Estimator::AbsorbError()", "xivo::hip::Feature::FillJacobianBlock(Eigen::Matrix<double, -1, -1"):
        assert name in syms, name
