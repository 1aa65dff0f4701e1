// mpmc_rotation.hpp -- rigid-body rotation helpers of the Monte Carlo drivers (mpmc_pimc.hpp, mpmc_gibbs.hpp).
//
// The drivers reproduce trajectories of the reference executable digit for digit, so the ARITHMETIC here follows the
// reference's rotation code operation by operation (src/Quaternion.cpp:20-127 for the rotor, src/Vector3D.h:96-113,131-138 and
// src/Vector3D.cpp:108-126 for the vector helpers): same association order, same degree -> radian constant.  The types and
// names are this repository's own.
#pragma once
#include <array>
#include <cmath>

namespace mpmc {

using Vec3 = std::array<double, 3>;

inline double dot3(const Vec3 &a, const Vec3 &b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline double length3(const Vec3 &a) { return std::sqrt(dot3(a, a)); }
inline Vec3 add3(const Vec3 &a, const Vec3 &b) { return Vec3{{a[0] + b[0], a[1] + b[1], a[2] + b[2]}}; }
inline Vec3 sub3(const Vec3 &a, const Vec3 &b) { return Vec3{{a[0] - b[0], a[1] - b[1], a[2] - b[2]}}; }
inline Vec3 scaled3(const Vec3 &a, double f) { return Vec3{{a[0] * f, a[1] * f, a[2] * f}}; }    // vector * scalar
inline Vec3 scaled3(double f, const Vec3 &a) { return Vec3{{f * a[0], f * a[1], f * a[2]}}; }    // scalar * vector
inline Vec3 divided3(const Vec3 &a, double f) { return Vec3{{a[0] / f, a[1] / f, a[2] / f}}; }
inline Vec3 cross3(const Vec3 &a, const Vec3 &b) {
	return Vec3{{a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}};
}
// unit vector of a; the zero vector stays zero
inline Vec3 unit3(const Vec3 &a) {
	const double len = length3(a);
	if (len != 0) return divided3(a, len);
	return Vec3{{0, 0, 0}};
}
// angle between two vectors, radians (NaN when rounding pushes the cosine past 1: callers inherit that from the reference)
inline double angle3(const Vec3 &a, const Vec3 &b) { return std::acos(dot3(a, b) / (length3(a) * length3(b))); }

// A rotation about an axis through the origin, held as the four numbers (v, s) = (axis sin(t/2), cos(t/2)).
struct Rotor {
	double vx, vy, vz, s;

	// rotation by `turn` radians about (ax, ay, az); a null axis gives the identity
	static Rotor about_axis(double ax, double ay, double az, double turn) {
		const double len = std::sqrt(ax * ax + ay * ay + az * az);
		if (len == 0.0) return Rotor{0, 0, 0, 1};
		ax = ax / len;
		ay = ay / len;
		az = az / len;
		const double half_sin = std::sin(turn / 2.0);
		return Rotor{ax * half_sin, ay * half_sin, az * half_sin, std::cos(turn / 2.0)};
	}
	// the same with the turn in degrees.  The divisor is the reference's constant (2.3e-10 short of 180 / pi): it is part of every
	// trajectory the goldens hold.
	static Rotor about_axis_degrees(double ax, double ay, double az, double turn_degrees) {
		return about_axis(ax, ay, az, turn_degrees / 57.2957795);
	}
	Rotor reversed() const { return Rotor{-vx, -vy, -vz, s}; }
	// Hamilton product (this, then nothing else: the plain product of the two quadruples)
	Rotor times(const Rotor &r) const {
		const double ps = s * r.s - vx * r.vx - vy * r.vy - vz * r.vz;
		const double px = s * r.vx + vx * r.s + vy * r.vz - vz * r.vy;
		const double py = s * r.vy - vx * r.vz + vy * r.s + vz * r.vx;
		const double pz = s * r.vz + vx * r.vy - vy * r.vx + vz * r.s;
		return Rotor{px, py, pz, ps};
	}
	static Rotor of_point(const double *p) { return Rotor{p[0], p[1], p[2], 0.0}; }
	// (R p) R~   -- the association of the reference's Quaternion::rotate (src/Quaternion.cpp:124-127)
	Vec3 turn_left_first(const Vec3 &p) const {
		const Rotor r = times(Rotor{p[0], p[1], p[2], 0.0}).times(reversed());
		return Vec3{{r.vx, r.vy, r.vz}};
	}
	// R (p R~)   -- the association of Molecule::rotate (src/Molecule.cpp:186-203)
	Vec3 turn_right_first(const Vec3 &p) const {
		const Rotor r = times(Rotor{p[0], p[1], p[2], 0.0}.times(reversed()));
		return Vec3{{r.vx, r.vy, r.vz}};
	}
};

} // namespace mpmc
