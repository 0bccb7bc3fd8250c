// Camera projection models with Jacobians, usable from device and host code.
//
// Restates the Project() member of the four reference cameras
// (/root/reference/common/camera_pinhole.h:17-37, camera_equidist.h:23-95,
// camera_radtan.h:23-90, camera_atan.h:22-67) behind the dispatcher
// CameraManager::Project (/root/reference/src/camera_manager.h:33-49).
// camera_project: xp and jac = d(xp)/d(xc). camera_project_jacc: additionally the intrinsics Jacobian `jacc` of the
// USE_ONLINE_CAMERA_CALIB builds (camera_pinhole.h:31-35, camera_atan.h:62-91, camera_radtan.h:78-96,
// camera_equidist.h:80-94), parameter order as each model's comment states it.
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

// d(xp)/d(intrinsics), 2 x dim row-major in jacc[2][9]; returns dim (4 pinhole, 5 atan, 9 radtan, 8 equidistant)
XIVO_HD int camera_project_jacc(const xivo_cam& c, double x, double y, double xp[2], double J[2][2], double jacc[2][9]) {
  camera_project(c, x, y, xp, J);
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 9; ++j) jacc[i][j] = 0.0;
  const double fx = c.fx, fy = c.fy;
  if (c.model == XIVO_CAM_PINHOLE) {          // d[x, y]_d[fx, fy, cx, cy]
    jacc[0][0] = x; jacc[0][2] = 1.0; jacc[1][1] = y; jacc[1][3] = 1.0;
    return 4;
  } else if (c.model == XIVO_CAM_EQUI) {      // d[x, y]_d[fx, fy, cx, cy, k0, k1, k2, k3]
    const double k0 = c.d[0], k1 = c.d[1], k2 = c.d[2], k3 = c.d[3];
    const double xy_norm = sqrt(x * x + y * y);
    const double th = atan2(xy_norm, 1.0), phi = atan2(y, x);
    const double th2 = th * th, th3 = th2 * th, th5 = th3 * th2, th7 = th5 * th2, th9 = th7 * th2;
    const double r = th + k0 * th3 + k1 * th5 + k2 * th7 + k3 * th9;
    const double cos_phi = cos(phi), sin_phi = sin(phi);
    jacc[0][0] = r * cos_phi; jacc[0][2] = 1.0; jacc[1][1] = r * sin_phi; jacc[1][3] = 1.0;
    const double dr_dk[4] = {th3, th5, th7, th9};
    for (int k = 0; k < 4; ++k) { jacc[0][4 + k] = fx * cos_phi * dr_dk[k]; jacc[1][4 + k] = fy * sin_phi * dr_dk[k]; }
    return 8;
  } else if (c.model == XIVO_CAM_RADTAN) {    // d[x, y]_d[fx, fy, cx, cy, p1, p2, k1, k2, k3]
    const double p1 = c.d[0], p2 = c.d[1], k1 = c.d[2], k2 = c.d[3], k3 = c.d[4];
    const double t2 = x * x, t3 = y * y;
    const double t6 = p1 * x * 2.0, t7 = p2 * y * 2.0, t8 = t2 * 3.0, t9 = t3 * 3.0;
    const double t10 = t6 * y, t11 = t7 * x, t12 = t2 + t3, t13 = t3 + t8, t14 = t2 + t9;
    const double t15 = t12 * t12, t16 = t12 * t12 * t12, t17 = k1 * t12;
    const double t18 = k2 * t15, t19 = k3 * t16, t20 = p1 * t14, t21 = p2 * t13;
    const double t28 = t17 + t18 + t19 + 1.0;
    const double t29 = t28 * x, t30 = t28 * y;
    const double t31 = t10 + t21 + t29, t32 = t11 + t20 + t30;
    jacc[0][0] = t31; jacc[0][2] = 1.0; jacc[0][4] = fx * x * y * 2.0; jacc[0][5] = fx * t13;
    jacc[0][6] = fx * t12 * x; jacc[0][7] = fx * t15 * x; jacc[0][8] = fx * t16 * x;
    jacc[1][1] = t32; jacc[1][3] = 1.0; jacc[1][4] = fy * t14; jacc[1][5] = fy * x * y * 2.0;
    jacc[1][6] = fy * t12 * y; jacc[1][7] = fy * t15 * y; jacc[1][8] = fy * t16 * y;
    return 9;
  } else {                                    // ATAN: d[x, y]_d[fx, fy, cx, cy, w]
    const double w = c.d[0];
    const double invw = 1.0 / w, w2 = 2.0 * tan(w * 0.5);
    const double R = sqrt(x * x + y * y);
    const bool singular = (R < 0.0001 || w == 0);
    if (singular) { jacc[0][0] = x; jacc[0][2] = 1.0; jacc[1][1] = y; jacc[1][3] = 1.0; return 5; }
    const double f = invw * atan(w2 * R) / R;
    jacc[0][0] = f * x; jacc[0][2] = 1.0; jacc[1][1] = f * y; jacc[1][3] = 1.0;
    const double df_dinvw = atan(w2 * R) / R, dinvw_dw = -invw * invw;
    const double df_datanw2R = invw / R, datanw2R_dw2R = 1 / (1 + (w2 * R) * (w2 * R)), dw2R_dw2 = R;
    double dw2_dw = 1 / cos(w * 0.5);
    dw2_dw *= dw2_dw;
    const double df_dw = df_dinvw * dinvw_dw + df_datanw2R * datanw2R_dw2R * dw2R_dw2 * dw2_dw;
    jacc[0][4] = fx * x * df_dw; jacc[1][4] = fy * y * df_dw;
    return 5;
  }
}

}  // namespace xivo_hip
