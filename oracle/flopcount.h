/* flopcount.h -- TEST INFRASTRUCTURE.  A drop-in `real` for the instrumented C++ build of myo_oracle.c (make flops ->
 * libmyo_oracle_flops.so): every floating-point add / subtract / multiply / divide / math call on a `real` bumps a
 * thread-local counter, so that stepping the oracle on seeded states yields the ALGORITHMIC flop count of the hot path
 * ("the figure of record is the flop count emitted by the oracle's instrumented build").
 * Convention: + - * /  = 1 flop each; sqrt counted on its own; sin cos asin acos exp pow = 1 "special" each; comparisons,
 * copies, negation, abs, floor and ceil are free.  The oracle's dense loops (efc_J rows over all nv, dense Cholesky of the
 * Newton Hessian) count what a sparsity-unaware implementation executes: the same convention MuJoCo's own dense paths give,
 * and an upper bound on what a structure-exploiting implementation (the HIP kernels) needs. */
#ifndef MYOO_FLOPCOUNT_H
#define MYOO_FLOPCOUNT_H
#include <cmath>
#include <cstdint>

struct FlopCounters { uint64_t add, mul, div, sqrt_, special; };
extern thread_local FlopCounters g_flops;

struct real {
  double v;
  real() = default;
  real(double x) : v(x) {}
  real(float x) : v(x) {}
  real(int x) : v(x) {}
  real(long x) : v((double)x) {}
  real(unsigned x) : v(x) {}
  real(unsigned long x) : v((double)x) {}
  explicit operator double() const { return v; }
  explicit operator float() const { return (float)v; }
  explicit operator int() const { return (int)v; }
  explicit operator long() const { return (long)v; }
  explicit operator bool() const { return v != 0; }
  real operator-() const { return real(-v); }
  real& operator+=(real o) { g_flops.add++; v += o.v; return *this; }
  real& operator-=(real o) { g_flops.add++; v -= o.v; return *this; }
  real& operator*=(real o) { g_flops.mul++; v *= o.v; return *this; }
  real& operator/=(real o) { g_flops.div++; v /= o.v; return *this; }
};
inline real operator+(real a, real b) { g_flops.add++; return real(a.v + b.v); }
inline real operator-(real a, real b) { g_flops.add++; return real(a.v - b.v); }
inline real operator*(real a, real b) { g_flops.mul++; return real(a.v * b.v); }
inline real operator/(real a, real b) { g_flops.div++; return real(a.v / b.v); }
#define MYOO_MIXED(T)                                                                                                    \
  inline real operator+(real a, T b) { return a + real(b); } inline real operator+(T a, real b) { return real(a) + b; } \
  inline real operator-(real a, T b) { return a - real(b); } inline real operator-(T a, real b) { return real(a) - b; } \
  inline real operator*(real a, T b) { return a * real(b); } inline real operator*(T a, real b) { return real(a) * b; } \
  inline real operator/(real a, T b) { return a / real(b); } inline real operator/(T a, real b) { return real(a) / b; } \
  inline bool operator<(real a, T b) { return a.v < b; } inline bool operator<(T a, real b) { return a < b.v; }         \
  inline bool operator>(real a, T b) { return a.v > b; } inline bool operator>(T a, real b) { return a > b.v; }         \
  inline bool operator<=(real a, T b) { return a.v <= b; } inline bool operator<=(T a, real b) { return a <= b.v; }     \
  inline bool operator>=(real a, T b) { return a.v >= b; } inline bool operator>=(T a, real b) { return a >= b.v; }     \
  inline bool operator==(real a, T b) { return a.v == b; } inline bool operator==(T a, real b) { return a == b.v; }     \
  inline bool operator!=(real a, T b) { return a.v != b; } inline bool operator!=(T a, real b) { return a != b.v; }
MYOO_MIXED(double)
MYOO_MIXED(float)
MYOO_MIXED(int)
#undef MYOO_MIXED
inline bool operator<(real a, real b) { return a.v < b.v; }
inline bool operator>(real a, real b) { return a.v > b.v; }
inline bool operator<=(real a, real b) { return a.v <= b.v; }
inline bool operator>=(real a, real b) { return a.v >= b.v; }
inline bool operator==(real a, real b) { return a.v == b.v; }
inline bool operator!=(real a, real b) { return a.v != b.v; }
inline bool operator!(real a) { return a.v == 0; }
inline real sqrt(real a) { g_flops.sqrt_++; return real(std::sqrt(a.v)); }
inline real fabs(real a) { return real(std::fabs(a.v)); }
inline real floor(real a) { return real(std::floor(a.v)); }
inline real ceil(real a) { return real(std::ceil(a.v)); }
#define MYOO_SPECIAL(f) inline real f(real a) { g_flops.special++; return real(std::f(a.v)); }
MYOO_SPECIAL(sin) MYOO_SPECIAL(cos) MYOO_SPECIAL(asin) MYOO_SPECIAL(acos) MYOO_SPECIAL(exp)
#undef MYOO_SPECIAL
inline real pow(real a, real b) { g_flops.special++; return real(std::pow(a.v, b.v)); }
inline real pow(real a, double b) { g_flops.special++; return real(std::pow(a.v, b)); }
inline real pow(real a, int b) { g_flops.special++; return real(std::pow(a.v, b)); }
#endif
