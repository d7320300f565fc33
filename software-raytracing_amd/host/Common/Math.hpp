// Math.hpp - the part of the reference's `math` namespace that the tracing path uses (Vec2/Vec3, Normalize, Dot,
// Cross, Reflect, Refract, Mix, Clamp, Mat3, GetRotationMat), written from scratch with the same names and
// semantics so that scene-building code written against Sources/Common/Math.hpp compiles unchanged.
// Semantics that matter (reference lines in /root/reference/Sources/Common/Math.hpp):
//   Normalize(0) == 0 (:233-238); Mat3 is built from COLUMNS and stored row-major (:401-406);
//   Reflect(v,n) = v - n*2*dot (:291-294); Refract falls back to Reflect on total internal reflection (:306-312);
//   GetRotationMat takes Euler angles in DEGREES and converts through a double pi (:913-930).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>

namespace math {

constexpr double kPi = 3.14159265358979323846;

template <typename T = float> struct Vec2 {
    T x{}, y{};
    Vec2() = default;
    Vec2(const T& a, const T& b) : x(a), y(b) {}
    Vec2 operator+(const Vec2& o) const { return {x + o.x, y + o.y}; }
    Vec2 operator-(const Vec2& o) const { return {x - o.x, y - o.y}; }
    Vec2 operator*(const T& s) const { return {x * s, y * s}; }
    Vec2 operator/(const T& s) const { return {x / s, y / s}; }
    Vec2 operator-() const { return {-x, -y}; }
};

template <typename T = float> struct Vec3 {
    union { T x, r; }; union { T y, g; }; union { T z, b; };
    Vec3() : x(0), y(0), z(0) {}
    Vec3(const T& a, const T& b_, const T& c) : x(a), y(b_), z(c) {}
    Vec3 operator+(const Vec3& o) const { return {x + o.x, y + o.y, z + o.z}; }
    Vec3 operator-(const Vec3& o) const { return {x - o.x, y - o.y, z - o.z}; }
    Vec3 operator*(const Vec3& o) const { return {x * o.x, y * o.y, z * o.z}; }
    Vec3 operator*(const T& s) const { return {x * s, y * s, z * s}; }
    Vec3 operator/(const T& s) const { return {x / s, y / s, z / s}; }
    Vec3 operator-() const { return {-x, -y, -z}; }
};

template <typename T> float LengthSquared(const Vec3<T>& v) { return static_cast<float>(v.x * v.x + v.y * v.y + v.z * v.z); }
template <typename T> float Length(const Vec3<T>& v) { return std::sqrt(LengthSquared(v)); }
template <typename T> Vec3<T> Normalize(const Vec3<T>& v) { const float l = Length(v); return l > 0 ? Vec3<T>{v.x / l, v.y / l, v.z / l} : Vec3<T>{}; }
template <typename T> T Dot(const Vec3<T>& a, const Vec3<T>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> Vec3<T> Cross(const Vec3<T>& a, const Vec3<T>& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
template <typename T> Vec3<T> Reflect(const Vec3<T>& v, const Vec3<T>& n) { return v - n * 2.0f * Dot(v, n); }
template <typename T> Vec3<T> Refract(const Vec3<T>& v, const Vec3<T>& n, float eta, bool reflectOnImpossibleAngles = true) {
    const T c = Dot(v, n);
    const T k = 1.0f - eta * eta * (1.0f - c * c);
    if (k < 0.0f) return reflectOnImpossibleAngles ? Reflect(v, n) : Vec3<T>{};
    return v * eta - n * (eta * c + std::sqrt(k));
}
template <typename T, typename R = float> T Mix(const T& a, const T& b, R ratio) { return a + (b - a) * ratio; }
template <typename T> T Clamp(const T& v, const T& lo, const T& hi) { return v < lo ? lo : (hi < v ? hi : v); }

template <typename T = float> struct Mat3 {
    T data[9] = {};
    Mat3() = default;
    explicit Mat3(T d) { data[0] = data[4] = data[8] = d; }
    Mat3(const Vec3<T>& i, const Vec3<T>& j, const Vec3<T>& k) {   // columns
        data[0] = i.x; data[1] = j.x; data[2] = k.x; data[3] = i.y; data[4] = j.y; data[5] = k.y; data[6] = i.z; data[7] = j.z; data[8] = k.z;
    }
    const T* row(size_t r) const { return data + 3 * r; }
    T* operator[](size_t r) { return data + 3 * r; }
    Vec3<T> operator*(const Vec3<T>& v) const {
        return {data[0] * v.x + data[1] * v.y + data[2] * v.z, data[3] * v.x + data[4] * v.y + data[5] * v.z, data[6] * v.x + data[7] * v.y + data[8] * v.z};
    }
};

template <typename T> T Determinant(const Mat3<T>& a) {
    const T* m = a.data;
    return (m[0] * m[4] * m[8]) + (m[1] * m[5] * m[6]) + (m[2] * m[3] * m[7]) - (m[2] * m[4] * m[6]) - (m[0] * m[5] * m[7]) - (m[1] * m[3] * m[8]);
}
// adjugate / determinant; the zero matrix when singular (the reference does the same, Math.hpp:635-654)
template <typename T> Mat3<T> Inverse(const Mat3<T>& a) {
    Mat3<T> r;
    const T det = Determinant(a);
    if (det == 0) return r;
    const T* m = a.data;
    const T s = 1 / det;
    r.data[0] = (m[4] * m[8] - m[7] * m[5]) * s; r.data[1] = (m[2] * m[7] - m[1] * m[8]) * s; r.data[2] = (m[1] * m[5] - m[2] * m[4]) * s;
    r.data[3] = (m[5] * m[6] - m[3] * m[8]) * s; r.data[4] = (m[0] * m[8] - m[2] * m[6]) * s; r.data[5] = (m[3] * m[2] - m[0] * m[5]) * s;
    r.data[6] = (m[3] * m[7] - m[6] * m[4]) * s; r.data[7] = (m[6] * m[1] - m[0] * m[7]) * s; r.data[8] = (m[0] * m[4] - m[3] * m[1]) * s;
    return r;
}

template <typename T = float> Mat3<T> GetRotationMat(const Vec3<T>& degrees) {
    const float ax = static_cast<float>(degrees.x * (kPi / 180.0f)), ay = static_cast<float>(degrees.y * (kPi / 180.0f)), az = static_cast<float>(degrees.z * (kPi / 180.0f));
    const float cx = std::cos(ax), sx = std::sin(ax), cy = std::cos(ay), sy = std::sin(ay), cz = std::cos(az), sz = std::sin(az);
    return Mat3<T>({cz * cy, sz * cy, -sy}, {cz * sy * sx - sz * cx, sz * sy * sx + cz * cx, cy * sx}, {cz * sy * cx + sz * sx, sz * sy * cx - cz * sx, cy * cx});
}

} // namespace math
