// earth.h -- batch descriptor of the Earth-absorption weight kernels (earth.hip)
#pragma once
#include <hip/hip_runtime.h>

namespace nrhip {

constexpr int EARTH_MAX_LAYERS = 16;

struct EarthModelDev {           // by value in the kernel arguments
    int n_layers;
    double earth_radius;
    double radii[EARTH_MAX_LAYERS];     // upper radius of layer k; layer k covers radii[k-1] <= r < radii[k]
    double coef[EARTH_MAX_LAYERS][4];   // rho(x) = ((c0 + c1 x) + c2 x^2) + c3 x^3, x = r / earth_radius
};

struct EarthBatch {
    long n;
    const double* zenith;        // [n]
    const double* energy;        // [n]
    const int* flavor;           // [n] PDG code, < 0 antiparticle
    const double* endpoint;      // [n][3] vertex, surface-centred (z < 0 below the surface)      (chord modes)
    const double* direction;     // [n][3] spherical_to_cartesian(zenith, azimuth), not normalised (chord modes)
    int mode;                    // NRHIP_EARTH_*
    int cross_section_type;      // NRHIP_XS_*
    double step;                 // integration step of the chord (500 m)
    double nucleon_mass;         // constants.m_p * units.kg
    double amu;                  // earth_attenuation.AMU
    double simple_radius, simple_density;          // get_simple_weight constants
    double layer_radii[3], layer_density[3];       // get_core_mantle_crust_weight constants
    double layer_theta[2];                         // pi - arcsin(radii[1] / radii[2]), pi - arcsin(radii[0] / radii[2])
};

void launch_earth_weights(hipStream_t s, const EarthBatch& b, const EarthModelDev& model, double* weight, double* slant_depth);

}  // namespace nrhip
