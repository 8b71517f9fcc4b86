// vecmath.hpp — host-side f32 vector/matrix arithmetic of the render path.
// Mirrors the operation ORDER of raytracer_lib/src/vecmath.rs (dot :74-76, cross :79-85,
// normalized :23-26, Matrix*Vec4 :200-211, Matrix*Matrix :237-313, rot_x/rot_y/translate
// :116-139, transpose :141-159) so that host-computed values (normals, camera matrices) carry
// the same f32 roundings as the reference.  Built with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstring>

namespace mi355rt {

struct Vec3 {
    float x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    Vec3 normalized() const
    {
        float len = std::sqrt(x * x + y * y + z * z);
        return Vec3(x / len, y / len, z / len);
    }
};
inline Vec3 operator+(const Vec3& a, const Vec3& b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator-(const Vec3& a, const Vec3& b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 operator*(const Vec3& a, float s) { return Vec3(a.x * s, a.y * s, a.z * s); }
inline Vec3 operator*(float s, const Vec3& a) { return Vec3(s * a.x, s * a.y, s * a.z); }
inline float dot(const Vec3& a, const Vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(const Vec3& a, const Vec3& b)
{
    return Vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

struct Vec4 {
    float x = 0, y = 0, z = 0, w = 0;
    Vec4() = default;
    Vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
    static Vec4 from_vec3(const Vec3& v) { return Vec4(v.x, v.y, v.z, 1.0f); }
    Vec3 xyz() const { return Vec3(x, y, z); }
};

// Row-major storage, row-vector convention: translation lives in e[12..14].
struct Matrix {
    float e[16];
    static Matrix ident()
    {
        Matrix m;
        std::memset(m.e, 0, sizeof m.e);
        m.e[0] = m.e[5] = m.e[10] = m.e[15] = 1.0f;
        return m;
    }
    static Matrix from_array(const float* a)
    {
        Matrix m;
        std::memcpy(m.e, a, sizeof m.e);
        return m;
    }
    static Matrix rot_x(float rad)
    {
        Matrix m = ident();
        m.e[5] = std::cos(rad); m.e[6] = -std::sin(rad); m.e[9] = std::sin(rad); m.e[10] = std::cos(rad);
        return m;
    }
    static Matrix rot_y(float rad)
    {
        Matrix m = ident();
        m.e[0] = std::cos(rad); m.e[2] = std::sin(rad); m.e[8] = -std::sin(rad); m.e[10] = std::cos(rad);
        return m;
    }
    static Matrix translate(const Vec3& v)
    {
        Matrix m = ident();
        m.e[12] = v.x; m.e[13] = v.y; m.e[14] = v.z;
        return m;
    }
    Matrix transpose() const
    {
        Matrix m = *this;
        m.e[1] = e[4];  m.e[2] = e[8];  m.e[3] = e[12];
        m.e[4] = e[1];  m.e[6] = e[9];  m.e[7] = e[13];
        m.e[8] = e[2];  m.e[9] = e[6];  m.e[11] = e[14];
        m.e[12] = e[3]; m.e[13] = e[7]; m.e[14] = e[11];
        return m;
    }
    bool operator==(const Matrix& o) const
    {
        for (int i = 0; i < 16; ++i) if (!(e[i] == o.e[i])) return false;
        return true;
    }
};
inline Vec4 operator*(const Matrix& m, const Vec4& v)
{
    return Vec4(v.x * m.e[0] + v.y * m.e[4] + v.z * m.e[8] + v.w * m.e[12],
                v.x * m.e[1] + v.y * m.e[5] + v.z * m.e[9] + v.w * m.e[13],
                v.x * m.e[2] + v.y * m.e[6] + v.z * m.e[10] + v.w * m.e[14],
                v.x * m.e[3] + v.y * m.e[7] + v.z * m.e[11] + v.w * m.e[15]);
}
inline Matrix operator*(const Matrix& s, const Matrix& r)
{
    Matrix o;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            o.e[4 * i + j] = s.e[4 * i] * r.e[j] + s.e[4 * i + 1] * r.e[4 + j]
                           + s.e[4 * i + 2] * r.e[8 + j] + s.e[4 * i + 3] * r.e[12 + j];
    return o;
}

}  // namespace mi355rt
