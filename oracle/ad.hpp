// TEST INFRASTRUCTURE (oracle) -- not part of the shipped product path.
// Minimal forward-mode automatic differentiation scalars used by the CPU oracle to obtain
// exact Jacobians/Hessians of the restated reference functions:
//   D1<N>  : value + N first-order tangents (vector forward mode)
//   DD<S>  : single-direction dual number over an arbitrary inner scalar S
// Hessian column k of f is read from f(DD<D1<N>>) seeded with outer direction e_k.
#pragma once
#include <cmath>

namespace orc {

inline double val(double a) { return a; }   // (declared first: DD<double> resolves val(a.v) at its definition)

template <int N>
struct D1 {
  double v;
  double g[N];
  D1() : v(0.0) { for (int i = 0; i < N; ++i) g[i] = 0.0; }
  D1(double c) : v(c) { for (int i = 0; i < N; ++i) g[i] = 0.0; }
  static D1 var(double c, int k) { D1 r(c); r.g[k] = 1.0; return r; }
};

template <int N> inline D1<N> operator+(const D1<N>& a, const D1<N>& b) { D1<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.g[i] = a.g[i] + b.g[i]; return r; }
template <int N> inline D1<N> operator-(const D1<N>& a, const D1<N>& b) { D1<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.g[i] = a.g[i] - b.g[i]; return r; }
template <int N> inline D1<N> operator-(const D1<N>& a) { D1<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.g[i] = -a.g[i]; return r; }
template <int N> inline D1<N> operator*(const D1<N>& a, const D1<N>& b) { D1<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.g[i] = a.g[i] * b.v + a.v * b.g[i]; return r; }
template <int N> inline D1<N> operator/(const D1<N>& a, const D1<N>& b) { D1<N> r; double ib = 1.0 / b.v; r.v = a.v * ib; for (int i = 0; i < N; ++i) r.g[i] = (a.g[i] - r.v * b.g[i]) * ib; return r; }
template <int N> inline D1<N> operator+(const D1<N>& a, double b) { D1<N> r = a; r.v += b; return r; }
template <int N> inline D1<N> operator+(double b, const D1<N>& a) { D1<N> r = a; r.v += b; return r; }
template <int N> inline D1<N> operator-(const D1<N>& a, double b) { D1<N> r = a; r.v -= b; return r; }
template <int N> inline D1<N> operator-(double b, const D1<N>& a) { D1<N> r = -a; r.v += b; return r; }
template <int N> inline D1<N> operator*(const D1<N>& a, double b) { D1<N> r; r.v = a.v * b; for (int i = 0; i < N; ++i) r.g[i] = a.g[i] * b; return r; }
template <int N> inline D1<N> operator*(double b, const D1<N>& a) { return a * b; }
template <int N> inline D1<N> operator/(const D1<N>& a, double b) { return a * (1.0 / b); }
template <int N> inline D1<N> operator/(double a, const D1<N>& b) { return D1<N>(a) / b; }
template <int N> inline D1<N>& operator+=(D1<N>& a, const D1<N>& b) { a = a + b; return a; }
template <int N> inline D1<N>& operator-=(D1<N>& a, const D1<N>& b) { a = a - b; return a; }
template <int N> inline D1<N>& operator*=(D1<N>& a, const D1<N>& b) { a = a * b; return a; }
template <int N> inline D1<N>& operator+=(D1<N>& a, double b) { a.v += b; return a; }
template <int N> inline D1<N> sin(const D1<N>& a) { D1<N> r; r.v = std::sin(a.v); double c = std::cos(a.v); for (int i = 0; i < N; ++i) r.g[i] = c * a.g[i]; return r; }
template <int N> inline D1<N> cos(const D1<N>& a) { D1<N> r; r.v = std::cos(a.v); double s = -std::sin(a.v); for (int i = 0; i < N; ++i) r.g[i] = s * a.g[i]; return r; }
template <int N> inline D1<N> sqrt(const D1<N>& a) { D1<N> r; r.v = std::sqrt(a.v); double s = 0.5 / r.v; for (int i = 0; i < N; ++i) r.g[i] = s * a.g[i]; return r; }
template <int N> inline double val(const D1<N>& a) { return a.v; }

template <class S>
struct DD {
  S v, d;
  DD() : v(0.0), d(0.0) {}
  DD(double c) : v(c), d(0.0) {}
  DD(const S& a, const S& b) : v(a), d(b) {}
};
template <class S> inline DD<S> operator+(const DD<S>& a, const DD<S>& b) { return DD<S>(a.v + b.v, a.d + b.d); }
template <class S> inline DD<S> operator-(const DD<S>& a, const DD<S>& b) { return DD<S>(a.v - b.v, a.d - b.d); }
template <class S> inline DD<S> operator-(const DD<S>& a) { return DD<S>(-a.v, -a.d); }
template <class S> inline DD<S> operator*(const DD<S>& a, const DD<S>& b) { return DD<S>(a.v * b.v, a.d * b.v + a.v * b.d); }
template <class S> inline DD<S> operator/(const DD<S>& a, const DD<S>& b) { S q = a.v / b.v; return DD<S>(q, (a.d - q * b.d) / b.v); }
template <class S> inline DD<S> operator+(const DD<S>& a, double b) { return DD<S>(a.v + b, a.d); }
template <class S> inline DD<S> operator+(double b, const DD<S>& a) { return DD<S>(a.v + b, a.d); }
template <class S> inline DD<S> operator-(const DD<S>& a, double b) { return DD<S>(a.v - b, a.d); }
template <class S> inline DD<S> operator-(double b, const DD<S>& a) { return DD<S>(b - a.v, -a.d); }
template <class S> inline DD<S> operator*(const DD<S>& a, double b) { return DD<S>(a.v * b, a.d * b); }
template <class S> inline DD<S> operator*(double b, const DD<S>& a) { return DD<S>(a.v * b, a.d * b); }
template <class S> inline DD<S> operator/(const DD<S>& a, double b) { return DD<S>(a.v / b, a.d / b); }
template <class S> inline DD<S> operator/(double a, const DD<S>& b) { return DD<S>(a) / b; }
template <class S> inline DD<S>& operator+=(DD<S>& a, const DD<S>& b) { a = a + b; return a; }
template <class S> inline DD<S>& operator-=(DD<S>& a, const DD<S>& b) { a = a - b; return a; }
template <class S> inline DD<S>& operator*=(DD<S>& a, const DD<S>& b) { a = a * b; return a; }
template <class S> inline DD<S>& operator+=(DD<S>& a, double b) { a.v += b; return a; }
template <class S> inline DD<S> sin(const DD<S>& a) { using std::sin; using std::cos; return DD<S>(sin(a.v), cos(a.v) * a.d); }
template <class S> inline DD<S> cos(const DD<S>& a) { using std::sin; using std::cos; return DD<S>(cos(a.v), -(sin(a.v) * a.d)); }
template <class S> inline DD<S> sqrt(const DD<S>& a) { using std::sqrt; S r = sqrt(a.v); return DD<S>(r, a.d / (r * 2.0)); }
template <class S> inline double val(const DD<S>& a) { return val(a.v); }

}  // namespace orc
