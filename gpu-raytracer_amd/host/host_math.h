// host_math.h — small float vector/matrix helpers for the host layer.
//
// The host layer restates the arithmetic the reference gets from glm 0.9.9.9 (vendored there under
// deps/include/glm, not copied here). Where operation ORDER matters for bit-equality with the
// reference's tables the formula is spelled out and the glm file it follows is cited.
// Build with -ffp-contract=off: the reference's x64 build does not contract a*b+c.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>

namespace okh {

struct Vec3 {
    float x = 0.f, y = 0.f, z = 0.f;
    Vec3() = default;
    Vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit Vec3(float s) : x(s), y(s), z(s) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 vmin(Vec3 a, Vec3 b) { return {b.x < a.x ? b.x : a.x, b.y < a.y ? b.y : a.y, b.z < a.z ? b.z : a.z}; }
inline Vec3 vmax(Vec3 a, Vec3 b) { return {a.x < b.x ? b.x : a.x, a.y < b.y ? b.y : a.y, a.z < b.z ? b.z : a.z}; }
inline float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline Vec3 normalize(Vec3 v) {  // glm/detail/func_geometric.inl: v * inversesqrt(dot(v,v))
    float inv = 1.f / std::sqrt(dot(v, v));
    return v * inv;
}

// Okay::AABB (reference source/Utilities.h:63-100)
struct AABB {
    Vec3 min{FLT_MAX, FLT_MAX, FLT_MAX};
    Vec3 max{-FLT_MAX, -FLT_MAX, -FLT_MAX};
    void growTo(Vec3 p) {
        min = vmin(p, min);
        max = vmax(p, max);
    }
    float getArea() const {  // "half area": xy + yz + zx (Utilities.h:78-82)
        Vec3 e = max - min;
        return e.x * e.y + e.y * e.z + e.z * e.x;
    }
    static bool intersects(const AABB& a, const AABB& b) {  // Utilities.h:84-90
        Vec3 ca = (a.max + a.min) * 0.5f, cb = (b.max + b.min) * 0.5f;
        Vec3 d{std::fabs(cb.x - ca.x), std::fabs(cb.y - ca.y), std::fabs(cb.z - ca.z)};
        Vec3 e = (a.max - a.min) * 0.5f + (b.max - b.min) * 0.5f;
        return d.x < e.x && d.y < e.y && d.z < e.z;
    }
};

// Column-major 4x4 like glm::mat4: c[col][row].
struct Mat4 {
    float c[4][4];
    static Mat4 identity() {
        Mat4 m;
        std::memset(&m, 0, sizeof(m));
        m.c[0][0] = m.c[1][1] = m.c[2][2] = m.c[3][3] = 1.f;
        return m;
    }
};
inline Mat4 transpose(const Mat4& m) {
    Mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.c[i][j] = m.c[j][i];
    return r;
}
// glm/detail/type_mat4x4.inl operator*(mat4,mat4): column j = A0*b0j + A1*b1j + A2*b2j + A3*b3j,
// summed left to right.
inline Mat4 mul(const Mat4& a, const Mat4& b) {
    Mat4 r;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            r.c[j][i] = ((a.c[0][i] * b.c[j][0] + a.c[1][i] * b.c[j][1]) + a.c[2][i] * b.c[j][2]) + a.c[3][i] * b.c[j][3];
    return r;
}
// glm operator*(mat4, vec4): (m0*x + m1*y) + (m2*z + m3*w)
inline void mulPoint(const Mat4& m, Vec3 v, float w, float out[4]) {
    for (int i = 0; i < 4; ++i) out[i] = (m.c[0][i] * v.x + m.c[1][i] * v.y) + (m.c[2][i] * v.z + m.c[3][i] * w);
}
// glm/detail/func_matrix.inl compute_inverse<4,4>: cofactor expansion, then * (1/det).
inline Mat4 inverse(const Mat4& m) {
    float Coef00 = m.c[2][2] * m.c[3][3] - m.c[3][2] * m.c[2][3];
    float Coef02 = m.c[1][2] * m.c[3][3] - m.c[3][2] * m.c[1][3];
    float Coef03 = m.c[1][2] * m.c[2][3] - m.c[2][2] * m.c[1][3];
    float Coef04 = m.c[2][1] * m.c[3][3] - m.c[3][1] * m.c[2][3];
    float Coef06 = m.c[1][1] * m.c[3][3] - m.c[3][1] * m.c[1][3];
    float Coef07 = m.c[1][1] * m.c[2][3] - m.c[2][1] * m.c[1][3];
    float Coef08 = m.c[2][1] * m.c[3][2] - m.c[3][1] * m.c[2][2];
    float Coef10 = m.c[1][1] * m.c[3][2] - m.c[3][1] * m.c[1][2];
    float Coef11 = m.c[1][1] * m.c[2][2] - m.c[2][1] * m.c[1][2];
    float Coef12 = m.c[2][0] * m.c[3][3] - m.c[3][0] * m.c[2][3];
    float Coef14 = m.c[1][0] * m.c[3][3] - m.c[3][0] * m.c[1][3];
    float Coef15 = m.c[1][0] * m.c[2][3] - m.c[2][0] * m.c[1][3];
    float Coef16 = m.c[2][0] * m.c[3][2] - m.c[3][0] * m.c[2][2];
    float Coef18 = m.c[1][0] * m.c[3][2] - m.c[3][0] * m.c[1][2];
    float Coef19 = m.c[1][0] * m.c[2][2] - m.c[2][0] * m.c[1][2];
    float Coef20 = m.c[2][0] * m.c[3][1] - m.c[3][0] * m.c[2][1];
    float Coef22 = m.c[1][0] * m.c[3][1] - m.c[3][0] * m.c[1][1];
    float Coef23 = m.c[1][0] * m.c[2][1] - m.c[2][0] * m.c[1][1];
    float Fac0[4] = {Coef00, Coef00, Coef02, Coef03}, Fac1[4] = {Coef04, Coef04, Coef06, Coef07};
    float Fac2[4] = {Coef08, Coef08, Coef10, Coef11}, Fac3[4] = {Coef12, Coef12, Coef14, Coef15};
    float Fac4[4] = {Coef16, Coef16, Coef18, Coef19}, Fac5[4] = {Coef20, Coef20, Coef22, Coef23};
    float V0[4] = {m.c[1][0], m.c[0][0], m.c[0][0], m.c[0][0]}, V1[4] = {m.c[1][1], m.c[0][1], m.c[0][1], m.c[0][1]};
    float V2[4] = {m.c[1][2], m.c[0][2], m.c[0][2], m.c[0][2]}, V3[4] = {m.c[1][3], m.c[0][3], m.c[0][3], m.c[0][3]};
    float Inv0[4], Inv1[4], Inv2[4], Inv3[4];
    for (int i = 0; i < 4; ++i) {
        Inv0[i] = (V1[i] * Fac0[i] - V2[i] * Fac1[i]) + V3[i] * Fac2[i];
        Inv1[i] = (V0[i] * Fac0[i] - V2[i] * Fac3[i]) + V3[i] * Fac4[i];
        Inv2[i] = (V0[i] * Fac1[i] - V1[i] * Fac3[i]) + V3[i] * Fac5[i];
        Inv3[i] = (V0[i] * Fac2[i] - V1[i] * Fac4[i]) + V2[i] * Fac5[i];
    }
    const float SignA[4] = {+1, -1, +1, -1}, SignB[4] = {-1, +1, -1, +1};
    Mat4 Inverse;
    for (int i = 0; i < 4; ++i) {
        Inverse.c[0][i] = Inv0[i] * SignA[i];
        Inverse.c[1][i] = Inv1[i] * SignB[i];
        Inverse.c[2][i] = Inv2[i] * SignA[i];
        Inverse.c[3][i] = Inv3[i] * SignB[i];
    }
    float Row0[4] = {Inverse.c[0][0], Inverse.c[1][0], Inverse.c[2][0], Inverse.c[3][0]};
    float Dot1 = (m.c[0][0] * Row0[0] + m.c[0][1] * Row0[1]) + (m.c[0][2] * Row0[2] + m.c[0][3] * Row0[3]);
    float OneOverDeterminant = 1.f / Dot1;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) Inverse.c[j][i] *= OneOverDeterminant;
    return Inverse;
}

inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

// glm::quat(vec3 eulerAngles) (glm/detail/type_quat.inl) then toMat3 (glm/gtc/quaternion.inl mat3_cast).
struct Mat3 {
    float c[3][3];
};
inline Mat3 rotationFromEulerDeg(Vec3 rotDeg) {
    Vec3 e{radians(rotDeg.x) * 0.5f, radians(rotDeg.y) * 0.5f, radians(rotDeg.z) * 0.5f};
    Vec3 c{std::cos(e.x), std::cos(e.y), std::cos(e.z)}, s{std::sin(e.x), std::sin(e.y), std::sin(e.z)};
    float w = c.x * c.y * c.z + s.x * s.y * s.z;
    float x = s.x * c.y * c.z - c.x * s.y * s.z;
    float y = c.x * s.y * c.z + s.x * c.y * s.z;
    float z = c.x * c.y * s.z - s.x * s.y * c.z;
    float qxx = x * x, qyy = y * y, qzz = z * z, qxz = x * z, qxy = x * y, qyz = y * z, qwx = w * x, qwy = w * y, qwz = w * z;
    Mat3 R;
    R.c[0][0] = 1.f - 2.f * (qyy + qzz);
    R.c[0][1] = 2.f * (qxy + qwz);
    R.c[0][2] = 2.f * (qxz - qwy);
    R.c[1][0] = 2.f * (qxy - qwz);
    R.c[1][1] = 1.f - 2.f * (qxx + qzz);
    R.c[1][2] = 2.f * (qyz + qwx);
    R.c[2][0] = 2.f * (qxz + qwy);
    R.c[2][1] = 2.f * (qyz - qwx);
    R.c[2][2] = 1.f - 2.f * (qxx + qyy);
    return R;
}

// Transform (reference source/Scene/Components.h:7-24)
struct Transform {
    Vec3 position{0.f, 0.f, 0.f};
    Vec3 rotation{0.f, 0.f, 0.f};  // degrees
    Vec3 scale{1.f, 1.f, 1.f};
    // translate(I,pos) * toMat4(quat(radians(rot))) * scale(I,scale)
    Mat4 calculateMatrix() const {
        Mat4 T = Mat4::identity();  // glm::translate: m[3] = m[0]*v0 + m[1]*v1 + m[2]*v2 + m[3]
        T.c[3][0] = position.x;
        T.c[3][1] = position.y;
        T.c[3][2] = position.z;
        Mat3 R3 = rotationFromEulerDeg(rotation);
        Mat4 R = Mat4::identity();
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) R.c[j][i] = R3.c[j][i];
        Mat4 S = Mat4::identity();
        S.c[0][0] = scale.x;
        S.c[1][1] = scale.y;
        S.c[2][2] = scale.z;
        return mul(mul(T, R), S);
    }
    Vec3 getForwardVec() const { Mat3 R = rotationFromEulerDeg(rotation); return normalize(Vec3(R.c[2][0], R.c[2][1], R.c[2][2])); }
    Vec3 getRightVec() const { Mat3 R = rotationFromEulerDeg(rotation); return normalize(Vec3(R.c[0][0], R.c[0][1], R.c[0][2])); }
    Vec3 getUpVec() const { Mat3 R = rotationFromEulerDeg(rotation); return normalize(Vec3(R.c[1][0], R.c[1][1], R.c[1][2])); }
};

// glm::perspectiveFovLH, depth range [-1,1] (x64 builds define no GLM_FORCE_DEPTH_ZERO_TO_ONE):
// glm/ext/matrix_clip_space.inl perspectiveFovLH_NO.
inline Mat4 perspectiveFovLH(float fov, float width, float height, float zNear, float zFar) {
    float rad = fov;
    float h = std::cos(0.5f * rad) / std::sin(0.5f * rad);
    float w = h * height / width;
    Mat4 r;
    std::memset(&r, 0, sizeof(r));
    r.c[0][0] = w;
    r.c[1][1] = h;
    r.c[2][2] = (zFar + zNear) / (zFar - zNear);
    r.c[2][3] = 1.f;
    r.c[3][2] = -(2.f * zFar * zNear) / (zFar - zNear);
    return r;
}
// glm::lookAtLH (glm/ext/matrix_transform.inl)
inline Mat4 lookAtLH(Vec3 eye, Vec3 center, Vec3 up) {
    Vec3 f = normalize(center - eye);
    Vec3 s = normalize(cross(up, f));
    Vec3 u = cross(f, s);
    Mat4 r = Mat4::identity();
    r.c[0][0] = s.x; r.c[1][0] = s.y; r.c[2][0] = s.z;
    r.c[0][1] = u.x; r.c[1][1] = u.y; r.c[2][1] = u.z;
    r.c[0][2] = f.x; r.c[1][2] = f.y; r.c[2][2] = f.z;
    r.c[3][0] = -dot(s, eye);
    r.c[3][1] = -dot(u, eye);
    r.c[3][2] = -dot(f, eye);
    return r;
}

}  // namespace okh
