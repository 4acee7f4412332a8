// Host-side H1 kinematics helpers (see h1_host_model.cpp).
#pragma once
namespace h1host {
void forward_kinematics(const double* x, double (*Rw)[9], double (*pw)[3]);
void reference_kinematics(const double* x, double* com, double* ee);
void reference_com_velocity(const double* x, double* comvel);
void foot_clearance(const double* q, double* clr);
void gravity_compensation(const double* x, const double* g, double* u);
}  // namespace h1host
