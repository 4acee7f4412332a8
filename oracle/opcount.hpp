// TEST INFRASTRUCTURE (oracle) -- not part of the shipped product path.
// Counting scalar for the algorithmic flop figures of SURVEY.md 8(d): the oracle's restatement, already templated on its
// scalar for forward-mode AD (ad.hpp), is instantiated on `Cnt` and every arithmetic operation it performs is tallied.
// Convention: add / subtract / multiply / divide / sqrt = 1 flop each, sin / cos = 1 each (tallied separately as well),
// a * b + c = 2 (nothing is fused); negations, comparisons, copies and conversions from double are free.
#pragma once
#include <cmath>

// (a namespace of its own: argument-dependent lookup then finds sqrt / sin / cos / val for the templates of h1_dynamics.hpp
// without dragging the whole of namespace orc into the overload sets of the re-compiled cost header)
namespace orc_count {

struct OpTally {
  long add = 0, mul = 0, div = 0, sqrt_ = 0, trig = 0;
  long flops() const { return add + mul + div + sqrt_ + trig; }
  void reset() { *this = OpTally(); }
};
inline OpTally& op_tally() { static thread_local OpTally t; return t; }

struct Cnt {
  double v;
  Cnt() : v(0.0) {}
  Cnt(double x) : v(x) {}
  Cnt(int x) : v(x) {}
};
inline Cnt operator+(const Cnt& a, const Cnt& b) { ++op_tally().add; return Cnt(a.v + b.v); }
inline Cnt operator-(const Cnt& a, const Cnt& b) { ++op_tally().add; return Cnt(a.v - b.v); }
inline Cnt operator*(const Cnt& a, const Cnt& b) { ++op_tally().mul; return Cnt(a.v * b.v); }
inline Cnt operator/(const Cnt& a, const Cnt& b) { ++op_tally().div; return Cnt(a.v / b.v); }
inline Cnt operator-(const Cnt& a) { return Cnt(-a.v); }
#define ORC_CNT_MIXED(op)                                                                   \
  inline Cnt operator op(const Cnt& a, double b) { return a op Cnt(b); }                    \
  inline Cnt operator op(double a, const Cnt& b) { return Cnt(a) op b; }                    \
  inline Cnt operator op(const Cnt& a, int b) { return a op Cnt((double)b); }               \
  inline Cnt operator op(int a, const Cnt& b) { return Cnt((double)a) op b; }
ORC_CNT_MIXED(+) ORC_CNT_MIXED(-) ORC_CNT_MIXED(*) ORC_CNT_MIXED(/)
#undef ORC_CNT_MIXED
inline Cnt& operator+=(Cnt& a, const Cnt& b) { a = a + b; return a; }
inline Cnt& operator-=(Cnt& a, const Cnt& b) { a = a - b; return a; }
inline Cnt& operator*=(Cnt& a, const Cnt& b) { a = a * b; return a; }
inline Cnt& operator/=(Cnt& a, const Cnt& b) { a = a / b; return a; }
inline Cnt& operator+=(Cnt& a, double b) { a = a + Cnt(b); return a; }
inline Cnt& operator-=(Cnt& a, double b) { a = a - Cnt(b); return a; }
inline Cnt& operator*=(Cnt& a, double b) { a = a * Cnt(b); return a; }
#define ORC_CNT_CMP(op)                                                                     \
  inline bool operator op(const Cnt& a, const Cnt& b) { return a.v op b.v; }                \
  inline bool operator op(const Cnt& a, double b) { return a.v op b; }                      \
  inline bool operator op(double a, const Cnt& b) { return a op b.v; }
ORC_CNT_CMP(<) ORC_CNT_CMP(>) ORC_CNT_CMP(<=) ORC_CNT_CMP(>=) ORC_CNT_CMP(==) ORC_CNT_CMP(!=)
#undef ORC_CNT_CMP
inline Cnt sqrt(const Cnt& a) { ++op_tally().sqrt_; return Cnt(std::sqrt(a.v)); }
inline Cnt sin(const Cnt& a) { ++op_tally().trig; return Cnt(std::sin(a.v)); }
inline Cnt cos(const Cnt& a) { ++op_tally().trig; return Cnt(std::cos(a.v)); }
inline Cnt fabs(const Cnt& a) { return Cnt(std::fabs(a.v)); }
inline double val(const Cnt& a) { return a.v; }

// Zero-aware variants, for the parts of the Jacobian scheme whose operands are structurally sparse.  An implementation that
// knows the tree skips them (a tangent direction moves only the bodies below its hinge and loads only the hinges above it;
// a unit acceleration likewise); the generic restatement multiplies the zeros through.  Counting rule: an operation one of
// whose operands is an exact zero costs nothing when its result is that operand or zero (0 * b, a + 0, 0 / b).
//   CntZ: plain counting scalar with that rule (mass matrix from unit accelerations).
//   TanD: dual number whose VALUE part is free (the primal quantities are computed once, elsewhere) and whose tangent part is
//         counted with that rule (one tangent direction of the inverse dynamics / of the integrator).
struct CntZ {
  double v;
  CntZ() : v(0.0) {}
  CntZ(double x) : v(x) {}
};
inline CntZ operator+(const CntZ& a, const CntZ& b) { if (a.v != 0.0 && b.v != 0.0) ++op_tally().add; return CntZ(a.v + b.v); }
inline CntZ operator-(const CntZ& a, const CntZ& b) { if (a.v != 0.0 && b.v != 0.0) ++op_tally().add; return CntZ(a.v - b.v); }
inline CntZ operator*(const CntZ& a, const CntZ& b) { if (a.v != 0.0 && b.v != 0.0) ++op_tally().mul; return CntZ(a.v * b.v); }
inline CntZ operator/(const CntZ& a, const CntZ& b) { if (a.v != 0.0) ++op_tally().div; return CntZ(a.v / b.v); }
inline CntZ operator-(const CntZ& a) { return CntZ(-a.v); }
#define ORC_CNTZ_MIXED(op)                                                                  \
  inline CntZ operator op(const CntZ& a, double b) { return a op CntZ(b); }                 \
  inline CntZ operator op(double a, const CntZ& b) { return CntZ(a) op b; }
ORC_CNTZ_MIXED(+) ORC_CNTZ_MIXED(-) ORC_CNTZ_MIXED(*) ORC_CNTZ_MIXED(/)
#undef ORC_CNTZ_MIXED
inline CntZ& operator+=(CntZ& a, const CntZ& b) { a = a + b; return a; }
inline CntZ& operator-=(CntZ& a, const CntZ& b) { a = a - b; return a; }
inline CntZ sqrt(const CntZ& a) { ++op_tally().sqrt_; return CntZ(std::sqrt(a.v)); }
inline CntZ sin(const CntZ& a) { ++op_tally().trig; return CntZ(std::sin(a.v)); }
inline CntZ cos(const CntZ& a) { ++op_tally().trig; return CntZ(std::cos(a.v)); }
inline double val(const CntZ& a) { return a.v; }

struct TanD {
  double v, d;
  TanD() : v(0.0), d(0.0) {}
  TanD(double x) : v(x), d(0.0) {}
  TanD(double x, double t) : v(x), d(t) {}
};
inline void tz_add(double a, double b) { if (a != 0.0 && b != 0.0) ++op_tally().add; }
inline void tz_mul(double a) { if (a != 0.0) ++op_tally().mul; }
inline TanD operator+(const TanD& a, const TanD& b) { tz_add(a.d, b.d); return TanD(a.v + b.v, a.d + b.d); }
inline TanD operator-(const TanD& a, const TanD& b) { tz_add(a.d, b.d); return TanD(a.v - b.v, a.d - b.d); }
inline TanD operator-(const TanD& a) { return TanD(-a.v, -a.d); }
inline TanD operator*(const TanD& a, const TanD& b) {         // d = a.d b.v + a.v b.d
  const double t1 = a.d * b.v, t2 = a.v * b.d;
  if (a.d != 0.0 && b.v != 0.0) ++op_tally().mul;
  if (b.d != 0.0 && a.v != 0.0) ++op_tally().mul;
  tz_add(t1, t2);
  return TanD(a.v * b.v, t1 + t2);
}
inline TanD operator/(const TanD& a, const TanD& b) {         // q = a.v / b.v (free); d = (a.d - q b.d) / b.v
  const double q = a.v / b.v;
  const double t = q * b.d;
  if (b.d != 0.0 && q != 0.0) ++op_tally().mul;
  tz_add(a.d, t);
  const double num = a.d - t;
  if (num != 0.0) ++op_tally().div;
  return TanD(q, num / b.v);
}
#define ORC_TAND_MIXED(op)                                                                  \
  inline TanD operator op(const TanD& a, double b) { return a op TanD(b); }                 \
  inline TanD operator op(double a, const TanD& b) { return TanD(a) op b; }
ORC_TAND_MIXED(+) ORC_TAND_MIXED(-) ORC_TAND_MIXED(*) ORC_TAND_MIXED(/)
#undef ORC_TAND_MIXED
inline TanD& operator+=(TanD& a, const TanD& b) { a = a + b; return a; }
inline TanD& operator-=(TanD& a, const TanD& b) { a = a - b; return a; }
inline TanD& operator*=(TanD& a, const TanD& b) { a = a * b; return a; }
inline TanD sqrt(const TanD& a) { const double r = std::sqrt(a.v); if (a.d != 0.0) { ++op_tally().mul; } return TanD(r, a.d * (0.5 / r)); }   // 0.5 / r: primal, free
inline TanD sin(const TanD& a) { tz_mul(a.d); return TanD(std::sin(a.v), std::cos(a.v) * a.d); }
inline TanD cos(const TanD& a) { tz_mul(a.d); return TanD(std::cos(a.v), -std::sin(a.v) * a.d); }
inline double val(const TanD& a) { return a.v; }

}  // namespace orc_count
