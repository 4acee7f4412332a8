// Host-side H1 kinematics used to build references the way the reference's loader does
// (RobotUtils::loadReferences, reference src/common/robot_utils.cpp:369-403: whole-body CoM from
// MuJoCo masses, ankle body positions) and the gravity-compensation cold start
// (RobotUtils::computeGravComp, reference src/common/robot_utils.cpp:844-866, with the correct dof
// index -- SURVEY.md Appendix D #8).  World-frame formulation, MJCF constants.
#include <cmath>

#include "h1_model_data.h"
#include "h1_host_model.h"
#include "h1_foot_hull.h"

namespace h1host {

static void quat_R(const double* q, double* R) {
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}
static void mul33(const double* A, const double* B, double* C) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
static void mv(const double* A, const double* x, double* y) { for (int i = 0; i < 3; ++i) y[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2]; }

void forward_kinematics(const double* x, double (*Rw)[9], double (*pw)[3]) {
  quat_R(x + 3, Rw[0]);
  for (int k = 0; k < 3; ++k) pw[0][k] = x[k];
  for (int i = 1; i < H1_NB; ++i) {
    const int p = H1_PARENT[i], a = H1_AXIS[i];
    const double th = x[7 + i - 1], c = std::cos(th), s = std::sin(th);
    double Ra[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    const int b = (a + 1) % 3, d = (a + 2) % 3;
    Ra[3 * b + b] = c; Ra[3 * b + d] = -s; Ra[3 * d + b] = s; Ra[3 * d + d] = c;
    double Rj[9]; mul33(&H1_RFIX[i][0][0], Ra, Rj);
    mul33(Rw[p], Rj, Rw[i]);
    double t[3]; mv(Rw[p], H1_POS[i], t);
    for (int k = 0; k < 3; ++k) pw[i][k] = pw[p][k] + t[k];
  }
}

void reference_kinematics(const double* x, double* com, double* ee) {
  double Rw[H1_NB][9], pw[H1_NB][3];
  forward_kinematics(x, Rw, pw);
  double m = 0.0; com[0] = com[1] = com[2] = 0.0;
  for (int i = 0; i < H1_NB; ++i) {
    double c[3]; mv(Rw[i], H1_COM[i], c);
    for (int k = 0; k < 3; ++k) com[k] += H1_MASS[i] * (pw[i][k] + c[k]);
    m += H1_MASS[i];
  }
  for (int k = 0; k < 3; ++k) { com[k] /= m; ee[k] = pw[H1_EE_LEFT][k]; ee[3 + k] = pw[H1_EE_RIGHT][k]; }
}

// whole-body CoM velocity J_com(q) qvel (mj_jacSubtreeCom of the root times qvel, robot_utils.cpp:383-391).
// MuJoCo free-joint velocity convention: qvel[0:3] linear velocity of the pelvis origin in the world frame,
// qvel[3:6] angular velocity in the pelvis frame, qvel[6:] hinge rates.
void reference_com_velocity(const double* x, double* cv) {
  double Rw[H1_NB][9], pw[H1_NB][3];
  forward_kinematics(x, Rw, pw);
  const double* v = x + H1_NQ;
  double ww[3]; mv(Rw[0], v + 3, ww);   // base angular velocity in the world frame
  double m = 0.0; cv[0] = cv[1] = cv[2] = 0.0;
  for (int i = 0; i < H1_NB; ++i) {
    double c[3]; mv(Rw[i], H1_COM[i], c);
    const double ci[3] = {pw[i][0] + c[0], pw[i][1] + c[1], pw[i][2] + c[2]};
    const double d0[3] = {ci[0] - pw[0][0], ci[1] - pw[0][1], ci[2] - pw[0][2]};
    double vel[3] = {v[0] + ww[1] * d0[2] - ww[2] * d0[1], v[1] + ww[2] * d0[0] - ww[0] * d0[2], v[2] + ww[0] * d0[1] - ww[1] * d0[0]};
    for (int j = 1; j < H1_NB; ++j) {
      if (!(j <= i && H1_ANC[j - 1][i - 1])) continue;   // hinge j moves body i
      const double z[3] = {Rw[j][H1_AXIS[j]], Rw[j][3 + H1_AXIS[j]], Rw[j][6 + H1_AXIS[j]]};
      const double d[3] = {ci[0] - pw[j][0], ci[1] - pw[j][1], ci[2] - pw[j][2]};
      const double qd = v[6 + j - 1];
      vel[0] += qd * (z[1] * d[2] - z[2] * d[1]); vel[1] += qd * (z[2] * d[0] - z[0] * d[2]); vel[2] += qd * (z[0] * d[1] - z[1] * d[0]);
    }
    for (int k = 0; k < 3; ++k) cv[k] += H1_MASS[i] * vel[k];
    m += H1_MASS[i];
  }
  for (int k = 0; k < 3; ++k) cv[k] /= m;
}

// Height of the lowest point of each foot's collision hull above the floor plane z = 0 (negative: penetration).
// The reference's schedule tool (get_contacts.py:96-147) sets qpos, runs mj_forward and marks a foot as in stance
// when MuJoCo reports a contact between the floor and an ankle-link geom: with the default margin 0 that is
// exactly "the lowest hull vertex is below the plane".  q = qpos[26] (MuJoCo order).
void foot_clearance(const double* q, double* clr) {
  double x[H1_NX] = {0};
  for (int i = 0; i < H1_NQ; ++i) x[i] = q[i];
  double Rw[H1_NB][9], pw[H1_NB][3];
  forward_kinematics(x, Rw, pw);
  const int body[2] = {H1_EE_LEFT, H1_EE_RIGHT};
  for (int f = 0; f < 2; ++f) {
    const double* R = Rw[body[f]];
    double zmin = 1e300;
    for (int i = 0; i < H1_FOOT_HULL_N; ++i) {
      const double z = R[6] * H1_FOOT_HULL[i][0] + R[7] * H1_FOOT_HULL[i][1] + R[8] * H1_FOOT_HULL[i][2];
      if (z < zmin) zmin = z;
    }
    clr[f] = pw[body[f]][2] + zmin;
  }
}

// qfrc_bias[6+j] at zero velocity = minus the generalized gravity force on hinge j
void gravity_compensation(const double* x, const double* g, double* u) {
  double Rw[H1_NB][9], pw[H1_NB][3];
  forward_kinematics(x, Rw, pw);
  for (int j = 1; j < H1_NB; ++j) {
    const double z[3] = {Rw[j][H1_AXIS[j]], Rw[j][3 + H1_AXIS[j]], Rw[j][6 + H1_AXIS[j]]};
    double q = 0.0;
    for (int i = j; i < H1_NB; ++i) {
      if (!H1_ANC[j - 1][i - 1]) continue;
      double c[3]; mv(Rw[i], H1_COM[i], c);
      const double d[3] = {pw[i][0] + c[0] - pw[j][0], pw[i][1] + c[1] - pw[j][1], pw[i][2] + c[2] - pw[j][2]};
      const double zxd[3] = {z[1] * d[2] - z[2] * d[1], z[2] * d[0] - z[0] * d[2], z[0] * d[1] - z[1] * d[0]};
      q += H1_MASS[i] * (zxd[0] * g[0] + zxd[1] * g[1] + zxd[2] * g[2]);
    }
    u[j - 1] = -q;
  }
}

}  // namespace h1host
