// rpt_vector.h — host-side float3/float4 and relativistic matrix helpers.
//
// Mirrors the interface of the reference's Vector.h:10-31 (same function names and argument
// meaning) on the rpt_float4 POD from rpt_layout.h.  All arithmetic is fp32, evaluated in the
// source order of the reference so that the Object[] buffers it produces match.  As in the
// reference, a "float3" is a 16-byte float4 whose .w is 0 (Vector.h:7) and dot() is 4-component
// (Vector.cpp:65-67).
#pragma once
#include "../../../include/rpt_layout.h"

namespace rpt {

inline rpt_float3 make_float3(float x, float y, float z) { return rpt_float3{x, y, z, 0.0f}; }
inline rpt_float4 make_float4(float x, float y, float z, float w) { return rpt_float4{x, y, z, w}; }

float sqr_magnitude(const rpt_float3 v);
float magnitude(const rpt_float3 v);
rpt_float3 normalize(const rpt_float3 v);
rpt_float3 operator+(const rpt_float3 &a, const rpt_float3 &b);
rpt_float3 &operator+=(rpt_float3 &a, const rpt_float3 &b);
rpt_float3 operator-(const rpt_float3 &a, const rpt_float3 &b);
rpt_float3 operator-(const rpt_float3 &v);
rpt_float3 operator*(const rpt_float3 &v, const float &c);
rpt_float3 operator*(const float &c, const rpt_float3 &v);
rpt_float3 operator/(const rpt_float3 &v, const float &c);
float dot(const rpt_float4 &a, const rpt_float4 &b);
rpt_float3 cross(const rpt_float3 &a, const rpt_float3 &b);
rpt_float3 elementwise_min(const rpt_float3 &a, const rpt_float3 &b);
rpt_float3 elementwise_max(const rpt_float3 &a, const rpt_float3 &b);

bool calcInvM(rpt_object &object);
void TRS(rpt_object &object, rpt_float3 translation, float angle, rpt_float3 axis, rpt_float3 scale);
void Identity(rpt_float4 (&M)[4]);
void Lorentz(rpt_float4 (&M)[4], rpt_float3 v);
rpt_float3 AddVelocity(rpt_float3 const &v1, rpt_float3 const &v2);
void MatrixMultiplyLeft(rpt_float4 (&A)[4], rpt_float4 const (&B)[4]);
void MatrixMultiplyRight(rpt_float4 const (&A)[4], rpt_float4 (&B)[4]);
void setLorentzBoost(rpt_object &object, rpt_float3 v);

// A default-constructed Object as the reference's in-class initialisers leave it
// (Object.h:10-21), with every field the reference leaves uninitialised set to zero.
rpt_object defaultObject();

}  // namespace rpt
