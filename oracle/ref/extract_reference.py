#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY (oracle build step; nothing under oracle/ is product code).

Cuts the member functions of the hot path out of the reference tree WHERE IT LIES (/root/reference/src/*.cpp, core.h) at
build time and writes them, byte for byte, to oracle/_ref/extracted/*.inc - a directory that is git-ignored, so no
reference source ever enters this repository. oracle/ref/xivo_refx.cpp then compiles exactly that text as member functions
of a shim class that declares the members they touch (#define Estimator RefEstimator): for these functions oracle/_ref is
"the reference compiled here", not a retyping. The Estimator translation units themselves cannot be compiled (OpenCV via
common/utils.h:17, jsoncpp, glog), which is why the functions are cut out instead of the files being built.

Each piece is located by its signature (not by a line number that a reference update would shift) and cut at the matching
closing brace; the script prints file:first-last for the build log and fails loudly when a signature is not found."""
import os
import re
import sys

PIECES = [
    # (output name, file, regex of the first line, kind)
    ("core_index_state", "src/core.h", r"^enum Index : int \{", "until:^struct State \\{"),      # enum Index ... struct State {...};
    ("feature_status", "src/core.h", r"^enum class FeatureStatus : int \{", "block"),
    ("update_joseph_form", "src/estimator.cpp", r"^void Estimator::UpdateJosephForm\(\) \{", "block"),
    ("compose_motion", "src/estimator.cpp", r"^void Estimator::ComposeMotion\(", "block"),
    ("compute_motion_jacobian_at", "src/estimator.cpp", r"^void Estimator::ComputeMotionJacobianAt\(", "block"),
    ("rk4_step", "src/rk4.cpp", r"^void Estimator::RK4Step\(", "block"),
    ("prince_dormand_step", "src/princedormand.cpp", r"^number_t Estimator::PrinceDormandStep\(", "block"),
    ("mh_gating", "src/update.cpp", r"^std::vector<FeaturePtr> Estimator::MHGating\(\) \{", "block"),
    ("filter_update", "src/update.cpp", r"^void Estimator::FilterUpdate\(\)", "block"),
    ("absorb_error_vec", "src/estimator.cpp", r"^void Estimator::AbsorbError\(const VecX &err\) \{", "block"),
    ("absorb_error", "src/estimator.cpp", r"^void Estimator::AbsorbError\(\) \{", "block"),
    ("imu_state_plus", "src/imu.cpp", r"^void IMUState::operator\+=\(const Tangent &dX\) \{", "block"),
    ("so3xr3", "src/group.h", r"^struct SO3xR3 \{", "block"),
    ("jacobian_cache", "src/jac.h", r"^struct JacobianCache \{", "block"),
    ("feature_xc", "src/feature.cpp", r"^Vec3 Feature::Xc\(Mat3 \*J\) \{", "block"),
    ("compute_jacobian", "src/feature.cpp", r"^void Feature::ComputeJacobian\(", "block"),
    ("fill_jacobian_block", "src/feature.cpp", r"^void Feature::FillJacobianBlock\(", "block"),
    # round 5: the rest of the path - OOS / loop-closure row builders, depth sub-filter, integrator outer loops, Propagate, 1-pt RANSAC
    ("oos_jacobian_struct", "src/jac.h", r"^struct OOSJacobian \{", "block"),
    ("observation", "src/core.h", r"^struct Observation \{", "block"),
    ("subfilter_options", "src/options.h", r"^struct SubfilterOptions \{", "block"),
    ("feature_xs", "src/feature.cpp", r"^Vec3 Feature::Xs\(const SE3 &gbc, Mat3 \*J\) \{", "block"),
    ("compute_oos_jacobian", "src/oos.cpp", r"^int Feature::ComputeOOSJacobian\(", "block"),
    ("compute_oos_jacobian_internal", "src/oos.cpp", r"^void Feature::ComputeOOSJacobianInternal\(", "block"),
    ("compute_lc_jacobian", "src/oos.cpp", r"^void Feature::ComputeLCJacobian\(", "block"),
    ("subfilter_update", "src/feature.cpp", r"^void Feature::SubfilterUpdate\(", "block"),
    ("rk4", "src/rk4.cpp", r"^void Estimator::RK4\(", "block"),
    ("prince_dormand", "src/princedormand.cpp", r"^void Estimator::PrinceDormand\(", "block"),
    ("propagate", "src/estimator.cpp", r"^void Estimator::Propagate\(bool visual_meas\) \{", "block"),
    ("one_point_ransac", "src/update.cpp", r"^Estimator::OnePointRANSAC\(", "block"),   # (its return type sits on the line above)
    ("find_new_ref_group", "src/estimator.cpp", r"^GroupPtr Estimator::FindNewRefGroup\(", "block"),
    ("backup_state", "src/estimator.cpp", r"^void Estimator::BackupState\(", "block"),
    ("restore_state", "src/estimator.cpp", r"^void Estimator::RestoreState\(", "block"),
]


def cut_block(lines, start):
    """lines[start] opens a definition: returns the index of the line holding its matching closing brace (comments and
    string literals of these functions hold no unbalanced braces - checked by the compile that follows)."""
    depth, seen = 0, False
    for i in range(start, len(lines)):
        code = re.sub(r"//.*$", "", lines[i])
        for ch in code:
            if ch == "{":
                depth += 1; seen = True
            elif ch == "}":
                depth -= 1
        if seen and depth == 0:
            return i
    raise RuntimeError("unbalanced braces")


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "_ref", "extracted")
    os.makedirs(out, exist_ok=True)
    manifest = []
    for name, rel, first, kind in PIECES:
        lines = open(os.path.join(ref, rel)).read().split("\n")
        idx = [i for i, l in enumerate(lines) if re.search(first, l)]
        if len(idx) != 1:
            sys.exit(f"extract_reference: {rel}: expected exactly one line matching {first!r}, found {len(idx)}")
        a = idx[0]
        if kind == "block":
            b = cut_block(lines, a)
        else:   # "until:<regex>": from the first line through the block that the second regex opens
            second = kind.split(":", 1)[1]
            j = [i for i in range(a, len(lines)) if re.search(second, lines[i])]
            if not j:
                sys.exit(f"extract_reference: {rel}: no line matching {second!r} after line {a + 1}")
            b = cut_block(lines, j[0])
        text = "\n".join(lines[a:b + 1]) + "\n"
        with open(os.path.join(out, name + ".inc"), "w") as f:
            f.write(f"// cut verbatim from {rel}:{a + 1}-{b + 1} by oracle/ref/extract_reference.py - not tracked\n")
            f.write(text)
        manifest.append(f"{name}: {rel}:{a + 1}-{b + 1} ({b - a + 1} lines)")
    open(os.path.join(out, "MANIFEST.txt"), "w").write("\n".join(manifest) + "\n")
    print("\n".join(manifest))


if __name__ == "__main__":
    main()
