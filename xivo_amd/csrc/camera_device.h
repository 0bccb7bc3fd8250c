// Camera projection models with Jacobians, usable from device and host code.
//
// Restates the Project() member of the four reference cameras
// (/root/reference/common/camera_pinhole.h:17-37, camera_equidist.h:23-95,
// camera_radtan.h:23-90, camera_atan.h:22-67) behind the dispatcher
// CameraManager::Project (/root/reference/src/camera_manager.h:33-49).
// Only xp and jac = d(xp)/d(xc) are produced (the intrinsics Jacobian `jacc`
// exists only under USE_ONLINE_CAMERA_CALIB, compiled out by default).
#pragma once
#include <math.h>
#include "../../include/xivo_hip.h"

#if defined(__HIPCC__)
#define XIVO_HD __host__ __device__ inline
#else
#define XIVO_HD inline
#endif

namespace xivo_hip {

// xc = (x, y) normalised camera coordinates -> pixel xp; J = [du/dx du/dy; dv/dx dv/dy]
XIVO_HD void camera_project(const xivo_cam& c, double x, double y, double xp[2], double J[2][2]) {
  const double fx = c.fx, fy = c.fy, cx = c.cx, cy = c.cy;
  if (c.model == XIVO_CAM_PINHOLE) {
    xp[0] = fx * x + cx;
    xp[1] = fy * y + cy;
    J[0][0] = fx; J[0][1] = 0.0; J[1][0] = 0.0; J[1][1] = fy;
  } else if (c.model == XIVO_CAM_EQUI) {
    const double k0 = c.d[0], k1 = c.d[1], k2 = c.d[2], k3 = c.d[3];
    const double xy_norm2 = x * x + y * y;
    const double xy_norm = sqrt(xy_norm2);
    const double xyz_norm2 = xy_norm2 + 1;
    const double th = atan2(xy_norm, 1.0);
    const double phi = atan2(y, x);
    const double th2 = th * th, th3 = th2 * th, th4 = th3 * th, th5 = th3 * th2;
    const double th6 = th5 * th, th7 = th5 * th2, th8 = th7 * th, th9 = th7 * th2;
    const double r = th + k0 * th3 + k1 * th5 + k2 * th7 + k3 * th9;
    const double cos_phi = cos(phi), sin_phi = sin(phi);
    xp[0] = fx * r * cos_phi + cx;
    xp[1] = fy * r * sin_phi + cy;
    const double dphi_dx = -y / xy_norm2, dphi_dy = x / xy_norm2;
    const double dth_dx = x / xyz_norm2 / xy_norm, dth_dy = y / xyz_norm2 / xy_norm;
    const double dr_dth = 1 + k0 * 3 * th2 + k1 * 5 * th4 + k2 * 7 * th6 + k3 * 9 * th8;
    J[0][0] = fx * cos_phi * dr_dth * dth_dx - fx * r * sin_phi * dphi_dx;
    J[0][1] = fx * cos_phi * dr_dth * dth_dy - fx * r * sin_phi * dphi_dy;
    J[1][0] = fy * sin_phi * dr_dth * dth_dx + fy * r * cos_phi * dphi_dx;
    J[1][1] = fy * sin_phi * dr_dth * dth_dy + fy * r * cos_phi * dphi_dy;
  } else if (c.model == XIVO_CAM_RADTAN) {
    const double p1 = c.d[0], p2 = c.d[1], k1 = c.d[2], k2 = c.d[3], k3 = c.d[4];
    const double t2 = x * x, t3 = y * y;
    const double t4 = k1 * x * 2.0, t5 = k1 * y * 2.0, t6 = p1 * x * 2.0, t7 = p2 * y * 2.0;
    const double t8 = t2 * 3.0, t9 = t3 * 3.0;
    const double t10 = t6 * y, t11 = t7 * x, t12 = t2 + t3, t13 = t3 + t8, t14 = t2 + t9;
    const double t15 = t12 * t12, t16 = t12 * t12 * t12, t17 = k1 * t12;
    const double t22 = k2 * t12 * x * 4.0, t23 = k2 * t12 * y * 4.0;
    const double t18 = k2 * t15, t19 = k3 * t16, t20 = p1 * t14, t21 = p2 * t13;
    const double t24 = k3 * t15 * x * 6.0, t25 = k3 * t15 * y * 6.0;
    const double t26 = t4 + t22 + t24, t27 = t5 + t23 + t25;
    const double t28 = t17 + t18 + t19 + 1.0;
    const double t29 = t28 * x, t30 = t28 * y;
    const double t31 = t10 + t21 + t29, t32 = t11 + t20 + t30;
    xp[0] = cx + fx * t31;
    xp[1] = cy + fy * t32;
    J[0][0] = fx * (t28 + p2 * x * 6.0 + p1 * y * 2.0 + t26 * x);
    J[0][1] = fx * (t6 + t7 + t27 * x);
    J[1][0] = fy * (t6 + t7 + t26 * y);
    J[1][1] = fy * (t28 + p2 * x * 2.0 + p1 * y * 6.0 + t27 * y);
  } else {  // XIVO_CAM_ATAN
    const double w = c.d[0];
    const double invw = 1.0 / w, w2 = 2.0 * tan(w * 0.5);
    const double R = sqrt(x * x + y * y);
    double f = 1;
    const bool singular = (R < 0.0001 || w == 0);
    if (!singular) f = invw * atan(w2 * R) / R;
    xp[0] = fx * f * x + cx;
    xp[1] = fy * f * y + cy;
    if (singular) {
      J[0][0] = fx; J[0][1] = 0.0; J[1][0] = 0.0; J[1][1] = fy;
    } else {
      const double a = w2 * R;
      const double df_dR = invw * (1. / (1 + a * a) * a - atan(a)) / R / R;
      const double df_dx = df_dR * x / R, df_dy = df_dR * y / R;
      J[0][0] = fx * f + fx * x * df_dx;
      J[0][1] = fx * x * df_dy;
      J[1][0] = fy * y * df_dx;
      J[1][1] = fy * f + fy * y * df_dy;
    }
  }
}

}  // namespace xivo_hip
