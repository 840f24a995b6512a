"""The C++ layer's headers and sources only use members the REAL Eigen / PCL / Ceres have (the drop-in
surface is the reference's C++ classes on those libraries' types).  None of the three is installed here, so the check is
a compile-only one against tests/cpp/mock_real — an API-subset mock declaring real member names only, nothing defined:
every header and source of csrc/cpp is parsed with -DPPCR_NO_COMPAT_TYPES (compat.hpp then includes <Eigen/...>,
<pcl/...>, <ceres/ceres.h> instead of defining its stand-ins).  A control makes sure the mock really rejects the members
only the stand-ins used to have.  No GPU, nothing from /root/reference."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "probabilistic_point_clouds_registration_amd", "csrc", "cpp")
MOCK = os.path.join(ROOT, "tests", "cpp", "mock_real")
BASE = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-DPPCR_NO_COMPAT_TYPES", "-I", MOCK, "-I", os.path.join(ROOT, "include"),
        "-I", os.path.join(CPP, "include")]

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not installed")


def _parse(path_or_code, is_code=False):
    if is_code:
        return subprocess.run(BASE + ["-x", "c++", "-"], input=path_or_code, capture_output=True, text=True, timeout=300)
    return subprocess.run(BASE + ["-x", "c++", path_or_code], capture_output=True, text=True, timeout=300)


def test_every_header_and_source_parses_against_the_real_member_names():
    units = sorted(glob.glob(os.path.join(CPP, "include", "prob_point_cloud_registration", "*.h*")) +
                   glob.glob(os.path.join(CPP, "src", "*.cc")))
    assert len(units) >= 14
    for u in units:
        if os.path.basename(u) == "compat_impl.hpp":      # (the stand-ins' own out-of-line parts: not used in this mode)
            continue
        # (a header is parsed through an #include, as users do: "#pragma once in main file" is no defect of the header)
        r = _parse(f'#include "{u}"\n', is_code=True) if u.endswith((".h", ".hpp")) else _parse(u)
        assert r.returncode == 0, f"{u}:\n{r.stderr[-3000:]}"


def test_user_code_written_against_the_reference_compiles_in_both_modes():
    """What a caller of the reference writes — the callback protocol, ErrorTerm's loss wrapper handed on as a
    ceres::LossFunction *, transforms composed and read through linear() / translation() — in real-library mode (mock)
    and against the stand-ins."""
    code = r'''
#include "prob_point_cloud_registration/prob_point_cloud_registration.h"
#include "prob_point_cloud_registration/prob_point_cloud_registration_iteration.hpp"
#include "prob_point_cloud_registration/utilities.hpp"
using namespace prob_point_cloud_registration;
double use(pcl::PointCloud<pcl::PointXYZ>::Ptr a, pcl::PointCloud<pcl::PointXYZ>::Ptr b) {
    ProbPointCloudRegistrationParams params;
    ProbPointCloudRegistration reg(a, b, params);
    reg.align();
    Eigen::Affine3d T = reg.transformation() * Eigen::Affine3d::Identity();
    ErrorTerm term((*a)[0], (*b)[0]);
    term.updateWeight(0.5);
    ceres::LossFunction *loss = term.weight();          // error_term.hpp:45: what AddResidualBlock takes
    double rho[3];
    loss->Evaluate(2.0, rho);
    Eigen::Quaterniond q(T.rotation());
    pcl::PointCloud<pcl::PointXYZ> moved;
    pcl::transformPointCloud(*a, moved, T);
    return rho[0] + T.translation()(0) + T.linear()(1, 2) + q.w() + moved.size();
}
'''
    r = _parse(code, is_code=True)
    assert r.returncode == 0, r.stderr[-3000:]
    standin = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(CPP, "include")]
    r = subprocess.run(standin + ["-x", "c++", "-"], input=code, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.parametrize("snippet", [
    "Eigen::Affine3d a = Eigen::Affine3d::from_rows(T);",
    "Eigen::Affine3d a = Eigen::Affine3d::Identity(); a.R(0, 0) = T[0];",
    "Eigen::SparseMatrix<double, Eigen::RowMajor> m; m.assign_csr(0, 0, {}, {}, {});",
    "prob_point_cloud_registration::ErrorTerm e(pcl::PointXYZ(), pcl::PointXYZ()); (void)e.weight()->scale();",
])
def test_the_mock_rejects_members_only_the_old_stand_ins_had(snippet):
    code = ('#include "prob_point_cloud_registration/error_term.hpp"\n#include "prob_point_cloud_registration/adapters.hpp"\n'
            "void f(const double *T) { " + snippet + " (void)T; }\n")
    r = _parse(code, is_code=True)
    assert r.returncode != 0 and "error" in r.stderr
