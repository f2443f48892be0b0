// fp64 3-vectors with the evaluation order the reference gets from GLM 0.9.6
// (dot = (x*x' + y*y') + z*z', normalize(v) = v * (1/sqrt(dot(v,v))), see
// glm/detail/func_geometric.inl of that release).  Shared by host (g++) and device (hipcc)
// code; both must be compiled with FMA contraction off so that every product and sum rounds
// separately.
#pragma once
#include <math.h>

#define CPT_PI 3.14159265358979311600  // reference: src/RayTracer.h:5
#if defined(__HIPCC__)
#define CPT_HD __host__ __device__ __forceinline__
#else
#define CPT_HD inline
#endif

struct dvec3 {
    double x, y, z;
};

CPT_HD dvec3 dv(double x, double y, double z) { dvec3 r; r.x = x; r.y = y; r.z = z; return r; }
CPT_HD dvec3 dv(const double* p) { return dv(p[0], p[1], p[2]); }
CPT_HD dvec3 operator+(dvec3 a, dvec3 b) { return dv(a.x + b.x, a.y + b.y, a.z + b.z); }
CPT_HD dvec3 operator-(dvec3 a, dvec3 b) { return dv(a.x - b.x, a.y - b.y, a.z - b.z); }
CPT_HD dvec3 operator-(dvec3 a) { return dv(-a.x, -a.y, -a.z); }
CPT_HD dvec3 operator*(dvec3 a, dvec3 b) { return dv(a.x * b.x, a.y * b.y, a.z * b.z); }
CPT_HD dvec3 operator*(dvec3 a, double s) { return dv(a.x * s, a.y * s, a.z * s); }
CPT_HD dvec3 operator*(double s, dvec3 a) { return dv(s * a.x, s * a.y, s * a.z); }
CPT_HD dvec3 operator/(dvec3 a, double s) { return dv(a.x / s, a.y / s, a.z / s); }
CPT_HD double dot(dvec3 a, dvec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
CPT_HD dvec3 cross(dvec3 a, dvec3 b) {
    return dv(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
CPT_HD double length(dvec3 v) { return sqrt(dot(v, v)); }
CPT_HD dvec3 normalize(dvec3 v) { return v * (1.0 / sqrt(dot(v, v))); }
CPT_HD double axis(dvec3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }
