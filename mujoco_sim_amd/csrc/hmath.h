// hmath.h — small fp64 host math helpers for the model compiler / host glue.
// Quaternions are (w,x,y,z); 3x3 matrices are row-major.
#pragma once
#include <cmath>
#include <cstring>

namespace hm {

inline void zero(double* r, int n) { for (int i = 0; i < n; i++) r[i] = 0; }
inline void copy(double* r, const double* a, int n) { for (int i = 0; i < n; i++) r[i] = a[i]; }
inline double dot3(const double* a, const double* b) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }
inline void cross(double* r, const double* a, const double* b) {
  double x = a[1]*b[2] - a[2]*b[1], y = a[2]*b[0] - a[0]*b[2], z = a[0]*b[1] - a[1]*b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
inline double norm3(const double* a) { return std::sqrt(dot3(a, a)); }
inline double normalize3(double* a) {
  double n = norm3(a);
  if (n < 1e-15) { a[0] = 1; a[1] = 0; a[2] = 0; return n; }
  a[0] /= n; a[1] /= n; a[2] /= n; return n;
}
inline void normalize4(double* q) {
  double n = std::sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
  if (n < 1e-15) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  for (int i = 0; i < 4; i++) q[i] /= n;
}
inline void mulquat(double* r, const double* a, const double* b) {
  double w = a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3];
  double x = a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2];
  double y = a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1];
  double z = a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
inline void quat2mat(double* m, const double* q) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w*w + x*x - y*y - z*z; m[1] = 2*(x*y - w*z);         m[2] = 2*(x*z + w*y);
  m[3] = 2*(x*y + w*z);         m[4] = w*w - x*x + y*y - z*z; m[5] = 2*(y*z - w*x);
  m[6] = 2*(x*z - w*y);         m[7] = 2*(y*z + w*x);         m[8] = w*w - x*x - y*y + z*z;
}
inline void rotvec(double* r, const double* m, const double* v) {  // r = M v
  double x = m[0]*v[0] + m[1]*v[1] + m[2]*v[2];
  double y = m[3]*v[0] + m[4]*v[1] + m[5]*v[2];
  double z = m[6]*v[0] + m[7]*v[1] + m[8]*v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
inline void rotvecT(double* r, const double* m, const double* v) {  // r = M^T v
  double x = m[0]*v[0] + m[3]*v[1] + m[6]*v[2];
  double y = m[1]*v[0] + m[4]*v[1] + m[7]*v[2];
  double z = m[2]*v[0] + m[5]*v[1] + m[8]*v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
inline void axisangle2quat(double* q, const double* axis, double angle) {
  double s = std::sin(angle * 0.5);
  q[0] = std::cos(angle * 0.5); q[1] = axis[0]*s; q[2] = axis[1]*s; q[3] = axis[2]*s;
}
// symmetric 3x3 eigen-decomposition by cyclic Jacobi; A = V diag(e) V^T, V row-major columns = eigenvectors
inline void eig3(const double A[9], double e[3], double V[9]) {
  double a[9]; copy(a, A, 9);
  double v[9] = {1,0,0, 0,1,0, 0,0,1};
  for (int sweep = 0; sweep < 50; sweep++) {
    double off = std::fabs(a[1]) + std::fabs(a[2]) + std::fabs(a[5]);
    if (off < 1e-300) break;
    double scale = std::fabs(a[0]) + std::fabs(a[4]) + std::fabs(a[8]);
    if (off < 1e-16 * scale) break;
    for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
      double apq = a[3*p+q];
      if (std::fabs(apq) < 1e-300) continue;
      double theta = (a[3*q+q] - a[3*p+p]) / (2 * apq);
      double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta*theta + 1));
      double c = 1 / std::sqrt(t*t + 1), s = t * c;
      for (int k = 0; k < 3; k++) {  // A <- A J
        double akp = a[3*k+p], akq = a[3*k+q];
        a[3*k+p] = c*akp - s*akq; a[3*k+q] = s*akp + c*akq;
      }
      for (int k = 0; k < 3; k++) {  // A <- J^T A
        double apk = a[3*p+k], aqk = a[3*q+k];
        a[3*p+k] = c*apk - s*aqk; a[3*q+k] = s*apk + c*aqk;
      }
      for (int k = 0; k < 3; k++) {
        double vkp = v[3*k+p], vkq = v[3*k+q];
        v[3*k+p] = c*vkp - s*vkq; v[3*k+q] = s*vkp + c*vkq;
      }
    }
  }
  e[0] = a[0]; e[1] = a[4]; e[2] = a[8];
  copy(V, v, 9);
}
inline void mat2quat(double* q, const double* m) {
  double tr = m[0] + m[4] + m[8];
  if (tr > 0) {
    double s = std::sqrt(tr + 1.0) * 2;
    q[0] = 0.25 * s; q[1] = (m[7] - m[5]) / s; q[2] = (m[2] - m[6]) / s; q[3] = (m[3] - m[1]) / s;
  } else if (m[0] > m[4] && m[0] > m[8]) {
    double s = std::sqrt(1.0 + m[0] - m[4] - m[8]) * 2;
    q[0] = (m[7] - m[5]) / s; q[1] = 0.25 * s; q[2] = (m[1] + m[3]) / s; q[3] = (m[2] + m[6]) / s;
  } else if (m[4] > m[8]) {
    double s = std::sqrt(1.0 + m[4] - m[0] - m[8]) * 2;
    q[0] = (m[2] - m[6]) / s; q[1] = (m[1] + m[3]) / s; q[2] = 0.25 * s; q[3] = (m[5] + m[7]) / s;
  } else {
    double s = std::sqrt(1.0 + m[8] - m[0] - m[4]) * 2;
    q[0] = (m[3] - m[1]) / s; q[1] = (m[2] + m[6]) / s; q[2] = (m[5] + m[7]) / s; q[3] = 0.25 * s;
  }
  normalize4(q);
}

}  // namespace hm
