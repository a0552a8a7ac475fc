/*
 * mi_pt.h -- C ABI of the MI355X spectral path-tracing hot path.
 *
 * This is the one boundary the build introduces: host C++ (scene front end,
 * Integrator-shaped class) -> C ABI -> hand-written HIP kernels (gfx950).
 * It replaces, for `Integrator "path"`, the body of the reference's
 *     SamplerIntegrator::Render(const Scene&)       src/core/integrator.cpp:228-342
 *     PathIntegrator::Li(...)                       src/integrators/path.cpp:64-188
 * The reference has no FFI: its integrator contract is the C++ virtual
 *     class Integrator { virtual void Render(const Scene &scene) = 0; }
 *                                                   src/core/integrator.h:53-58
 * and Scene / BVHAccel keep their data private (src/core/scene.h:76-79,
 * src/accelerators/bvh.h:91-94), so the drop-in is at the `.pbrt` file +
 * `Integrator "path"` level: the host front end flattens the scene into the POD
 * description below and calls mi_pt_*.
 *
 * Conventions: plain pointers + counts, host memory unless a flag says device;
 * inputs are copied at mi_pt_create (caller keeps ownership); every entry point
 * returns 0 on success or a negative mi_status, never aborts, never throws;
 * one handle per GPU, calls on one handle are serialised by the caller.
 * Spectrum = float[MI_NSPEC] (31 bins over 395..705 nm, src/core/spectrum.h:48-50).
 */
#ifndef MI_PT_H
#define MI_PT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_NSPEC 31
#define MI_ABI_VERSION 10
#define MI_MAX_BXDFS 8 /* BSDF::MaxBxDFs, src/core/reflection.h:196 */

typedef enum mi_status {
    MI_OK = 0,
    MI_ERR_INVALID = -1,   /* bad argument / malformed description */
    MI_ERR_NO_DEVICE = -2, /* no HIP device / ordinal out of range */
    MI_ERR_HIP = -3,       /* a HIP runtime call failed (see mi_pt_last_error) */
    MI_ERR_UNSUPPORTED = -4,
    MI_ERR_NOMEM = -5
} mi_status;

/* ---- acceleration structure: same 32-byte node as LinearBVHNode
 *      (src/accelerators/bvh.cpp:95-104), depth-first, first child = idx+1. */
typedef struct mi_bvh_node {
    float bmin[3];
    float bmax[3];
    int32_t offset;    /* leaf: first primitive; interior: second child */
    uint16_t n_prims;  /* 0 -> interior */
    uint8_t axis;
    uint8_t pad;
} mi_bvh_node;

/* One GeometricPrimitive (src/core/primitive.h:70-95) in BVH leaf order. */
typedef struct mi_prim {
    int32_t shape;      /* >=0: triangle index; <0: ~sphere index (ignored when `instance` is set) */
    int32_t material;   /* index into materials, -1 = none (interface only) */
    int32_t area_light; /* index into lights, -1 = not emissive */
    int32_t instance;   /* 0: a GeometricPrimitive; k + 1: the TransformedPrimitive of mi_scene_desc.instances[k] (ABI v7) */
} mi_prim;

/* One ObjectInstance (TransformedPrimitive, src/core/primitive.cpp:78-99; api.cpp:1570-1615): the primitives of a named
 * object, kept in the space they were declared in behind a BVH of their own (`root`: that tree's root in
 * mi_scene_desc.nodes, whose interior / leaf offsets are absolute like the world tree's), reached by carrying the ray into
 * that space with WorldToInstance (Transform::operator()(Ray), transform.h:262-277: origin error bound, dt shift of the
 * origin and of tMax) and carrying the SurfaceInteraction back with InstanceToWorld (transform.cpp:262-297). Matrices row
 * major, m[r*4+c]; w2i is the stored inverse of i2w (Transform::mInv), not a recomputation. Objects hold no area lights. */
typedef struct mi_instance {
    float i2w[16], w2i[16];
    uint32_t root;
    uint32_t pad[3];
} mi_instance;

/* Per TriangleMesh flags (src/shapes/triangle.cpp:54-92). */
#define MI_MESH_HAS_N 1u
#define MI_MESH_HAS_UV 2u
#define MI_MESH_FLIP 4u /* reverseOrientation ^ transformSwapsHandedness */
typedef struct mi_mesh {
    uint32_t flags;
    uint32_t first_vertex, n_vertices;
    uint32_t first_tri, n_tris;
    /* "alpha" / "shadowalpha" float textures (triangle.cpp:331-338,531-570,716-740): index into textures, -1 = none.
     * A hit whose texture value is exactly 0 is no hit ("shadowalpha" for IntersectP only). The value is channel 0 of the
     * texture (float image textures are stored grey), looked up with zero footprint. */
    int32_t alpha_tex, shadow_alpha_tex;
} mi_mesh;

/* Sphere (src/shapes/sphere.h:49-66). Matrices row-major, m[r*4+c]. */
typedef struct mi_sphere {
    float o2w[16], w2o[16];
    float radius, z_min, z_max, theta_min, theta_max, phi_max;
    int32_t reverse_orientation, swaps_handedness;
} mi_sphere;

/* ---- materials: with constant textures (src/textures/constant.h:49-58) a
 * Material::ComputeScatteringFunctions result is a fixed BxDF list; image textures bind to lobes through mi_lobe_tex. */
typedef enum mi_bxdf_type {
    MI_BXDF_LAMBERTIAN_REFLECTION = 0, /* R */
    MI_BXDF_OREN_NAYAR,                /* R, p[0]=A, p[1]=B */
    MI_BXDF_SPECULAR_REFLECTION,       /* R, fresnel */
    MI_BXDF_SPECULAR_TRANSMISSION,     /* R(=T), p[0]=etaA, p[1]=etaB */
    MI_BXDF_FRESNEL_SPECULAR,          /* R, S(=T), p[0]=etaA, p[1]=etaB */
    MI_BXDF_MICROFACET_REFLECTION,     /* R, p[0]=alphax, p[1]=alphay, fresnel, p[5]=separableG */
    MI_BXDF_MICROFACET_TRANSMISSION,   /* R(=T), p[0]=alphax, p[1]=alphay, p[2]=etaA, p[3]=etaB, p[5]=separableG */
    MI_BXDF_LAMBERTIAN_TRANSMISSION,   /* R(=T) */
    MI_BXDF_DISNEY_DIFFUSE,            /* R */
    MI_BXDF_DISNEY_FAKE_SS,            /* R, p[0]=roughness */
    MI_BXDF_DISNEY_RETRO,              /* R, p[0]=roughness */
    MI_BXDF_DISNEY_SHEEN,              /* R */
    MI_BXDF_DISNEY_CLEARCOAT,          /* p[0]=weight, p[1]=gloss */
    MI_BXDF_FRESNEL_BLEND              /* R(=Rd), S(=Rs), p[0]=alphax, p[1]=alphay (reflection.cpp:279-298,450-475) */
} mi_bxdf_type;

typedef enum mi_fresnel_type {
    MI_FRESNEL_NOOP = 0,
    MI_FRESNEL_DIELECTRIC, /* p[2]=etaI, p[3]=etaT */
    MI_FRESNEL_DISNEY,     /* S=R0, p[2]=metallic, p[3]=eta */
    MI_FRESNEL_CONDUCTOR   /* etaI = 1, S=etaT, K=k per bin (FrConductor, reflection.cpp:71-94) */
} mi_fresnel_type;

/* BxDFType bits, src/core/reflection.h:153-161 */
#define MI_BSDF_REFLECTION 1
#define MI_BSDF_TRANSMISSION 2
#define MI_BSDF_DIFFUSE 4
#define MI_BSDF_GLOSSY 8
#define MI_BSDF_SPECULAR 16
#define MI_BSDF_ALL 31

typedef struct mi_bxdf {
    int32_t type;    /* mi_bxdf_type */
    int32_t flags;   /* BxDFType bits */
    int32_t fresnel; /* mi_fresnel_type (reflection lobes) */
    int32_t scaled;  /* the number of ScaledBxDF wrappers around the lobe (mix materials, reflection.cpp:96-107): 1: f = scale * f;
                        2 (ABI v10, a "mix" of a "mix"): f = scale2 * (scale * f) */
    float p[8];
    float R[MI_NSPEC];
    float S[MI_NSPEC];
    float K[MI_NSPEC];     /* conductor absorption k */
    float scale[MI_NSPEC]; /* ScaledBxDF::scale */
    float scale2[MI_NSPEC]; /* the outer ScaledBxDF's scale when scaled == 2 */
} mi_bxdf;

/* A lobe whose spectrum comes from an image texture (ABI v5). The lobe list of a material stays fixed; at a hit the
 * texture value T = Clamp(Spectrum::FromRGB(mipmap lookup)) (imagemap.h:82-93,113-117; FromRGB's default type is
 * Illuminant, spectrum.h:428-429, so the conversion uses rgb_illum like the environment light's) replaces R (or S), or scales
 * the constant stored there when MI_LOBE_TEX_MUL_* is set (uber: `op * Kd->Evaluate(si).Clamp()`, uber.cpp:71-72;
 * translucent: `r * kd`, translucent.cpp:62), and the lobe is left out of the BSDF at that hit when the spectrum the
 * material tests with IsBlack() is black (`if (!r.IsBlack()) bsdf->Add(...)`, matte.cpp:58-63). */
#define MI_LOBE_TEX_MUL_R 1u
#define MI_LOBE_TEX_MUL_S 2u
typedef enum mi_lobe_rule {
    MI_LOBE_IF_R = 0,      /* present iff the lobe's R (after the multiplier) is not black */
    MI_LOBE_IF_R_OR_S = 1, /* present iff R or S is not black (FresnelSpecular glass.cpp:70-72,78-80; FresnelBlend substrate.cpp:53) */
    MI_LOBE_IF_TEX = 2,    /* present iff the texture value itself is not black (translucent.cpp:59-60,68-69) */
    /* ABI v10 -- "disney" with an image-textured "color" (disney.cpp:485-587). The material adds its lobes whatever the colour
     * is, and three of their spectra are not linear in it. With c = the texture value (clamped), lum = c.y() and
     * Ctint = lum > 0 ? c / lum : 1 (per bin), the lobe's textured channel is:
     *   MI_LOBE_ALWAYS          R = c (or the constant R times c with MI_LOBE_TEX_MUL_R): the diffuse, retro, fake-subsurface and
     *                           diffuse-transmission lobes, whose weights are the constant
     *   MI_LOBE_DISNEY_SHEEN    R = Lerp(p[7], 1, Ctint) * p[6]            (Csheen * (diffuseWeight * sheenWeight))
     *   MI_LOBE_DISNEY_SPEC     R = c, S = Lerp(p[2], Lerp(p[6], 1, Ctint) * p[7], c)   (Cspec0; p[2] = metallic, p[6] = specularTint,
     *                           p[7] = SchlickR0FromEta(eta)) -- the microfacet lobe with the Disney Fresnel term
     *   MI_LOBE_DISNEY_STRANS   R = sqrt(c) * p[6]                          (strans * Sqrt(c))
     * where Lerp(t, a, b) = a * (1 - t) + b * t as Spectrum arithmetic does it (spectrum.h:577-580). */
    MI_LOBE_ALWAYS = 3,
    MI_LOBE_DISNEY_SHEEN = 4,
    MI_LOBE_DISNEY_SPEC = 5,
    MI_LOBE_DISNEY_STRANS = 6,
    /* "metal" with image-textured `eta` / `k` (metal.cpp:119-122: FresnelConductor(1, eta->Evaluate(si), k->Evaluate(si))):
     * tex_S = eta's texture, tex_R = k's -- the lobe's R stays the constant 1 -- either may be -1; always present */
    MI_LOBE_METAL = 7
} mi_lobe_rule;
typedef struct mi_lobe_tex {
    int32_t tex_R, tex_S; /* index into mi_scene_desc.textures, -1 = the constant in mi_bxdf */
    uint32_t flags;       /* MI_LOBE_TEX_MUL_* */
    int32_t rule;         /* mi_lobe_rule */
} mi_lobe_tex;

typedef struct mi_material {
    int32_t n_bxdfs;
    float eta; /* BSDF::eta, src/core/reflection.h:189 */
    int32_t kind; /* informational: 0 matte 1 plastic 2 glass 3 uber 4 disney 5 mirror 6 metal 7 substrate 8 translucent 9 mix */
    int32_t textured; /* != 0: some lobe has tex_R / tex_S >= 0, or bump_tex >= 0 */
    mi_bxdf bxdf[MI_MAX_BXDFS];
    mi_lobe_tex tex[MI_MAX_BXDFS];
    int32_t bump_tex; /* "bumpmap": float image texture displacing the shading geometry (Material::Bump, material.cpp:47-84), -1 = none */
    /* "roughness" / "uroughness" / "vroughness" given as float image textures (plastic.cpp:57-62, uber.cpp:88-96,
     * substrate.cpp:55-60, metal.cpp:66-73, translucent.cpp:70-72): texture of the u / v roughness of the material's
     * microfacet lobes, -1 = the constant alpha in mi_bxdf.p[0] / p[1]. The value at the hit, through RoughnessToAlpha when
     * MI_ROUGH_REMAP is set, replaces that constant in every microfacet / FresnelBlend lobe of the material. */
    int32_t rough_tex[2];
    uint32_t rough_flags;
    /* ABI v10 -- "matte" with `sigma` a float image texture (matte.cpp:55-62): at the hit sig = Clamp(value, 0, 90), and the
     * material's diffuse lobe (compiled as MI_BXDF_OREN_NAYAR) is a LambertianReflection where sig == 0 and an OrenNayar with
     * the A, B of that sig elsewhere (reflection.h:414-420). -1 = the constants of the lobe. */
    int32_t sigma_tex;
} mi_material;
#define MI_ROUGH_REMAP 1u
/* ABI v10 -- "glass" with `uroughness` / `vroughness` float image textures (glass.cpp:60-92): the material holds the lobes of both
 * branches -- lobe 0 the FresnelSpecular one, after it the microfacet reflection / transmission lobes -- and the hit decides:
 * `isSpecular = urough == 0 && vrough == 0` on the values before the remap (a constant axis keeps its raw value in lobe 0's
 * p[6] (u) / p[7] (v)); lobe 0 is part of the BSDF where isSpecular holds, the others where it does not. */
#define MI_ROUGH_GLASS 2u
/* ABI v10 -- "disney" with `roughness` a float image texture (disney.cpp:491, 538-541, 568-573): rough_tex[0] is the map and the
 * value at the hit, rough, replaces p[0] of the DISNEY_FAKE_SS / DISNEY_RETRO lobes and gives the microfacet lobes their alphas:
 * max(.001, sqr(r) / p[4]), max(.001, sqr(r) * p[4]) with p[4] = aspect (from "anisotropic") and r = rough -- or p[7] * rough for the
 * thin surface's transmission lobe, whose p[7] = 0.65 eta - 0.35 (0 elsewhere). */
#define MI_ROUGH_DISNEY 4u

/* ImageTexture<RGBSpectrum, Spectrum> with UVMapping2D (src/textures/imagemap.h, src/core/texture.cpp:91-99) over a
 * MIPMap<RGBSpectrum> (src/core/mipmap.h). The pyramid is built on the host exactly as the reference builds it (y flip,
 * scale, inverse gamma, power-of-two Lanczos resampling, 2x2 box levels); the render side filters it per hit with the
 * reference's EWA / trilinear lookup from the hit's (u, v) and its screen-space derivatives. */
#define MI_MAX_MIP_LEVELS 16
typedef struct mi_mipmap {
    int32_t n_levels;
    int32_t wrap;           /* 0 repeat, 1 black, 2 clamp (ImageWrap, mipmap.h:50) */
    int32_t width, height;  /* level 0; level l is max(1, width >> l) x max(1, height >> l) */
    const float *texels;    /* RGB triples, all levels back to back, rows from t = 0 */
    uint32_t level_offset[MI_MAX_MIP_LEVELS]; /* first texel of level l */
} mi_mipmap;
typedef enum mi_tex_filter { MI_TEX_EWA = 0, MI_TEX_TRILINEAR = 1, MI_TEX_NONE = 2 /* "noFiltering" */ } mi_tex_filter;
typedef struct mi_texture {   /* spectrum textures: RGB pyramid; float textures: the same with r = g = b (convertIn, imagemap.h:107-110) */
    int32_t mipmap;         /* index into mi_scene_desc.mipmaps */
    int32_t filter;         /* mi_tex_filter */
    float max_aniso;        /* "maxanisotropy" */
    float su, sv, du, dv;   /* UVMapping2D */
    float post_scale;       /* float textures: the looked-up value is multiplied by this (1 for a plain image texture; the
                             * constant operand of a Texture "scale" over an image texture, scale.h:56-58) */
    int32_t type;           /* mi_tex_type */
    int32_t aa_none;        /* checkerboard: "aamode" "none" (point sampled) instead of the closed-form box filter */
    float spec1[MI_NSPEC], spec2[MI_NSPEC]; /* checkerboard: "tex1" / "tex2" (constant spectra) */
} mi_texture;
/* MI_TEX_CHECKERBOARD: Checkerboard2DTexture<Spectrum> over UVMapping2D (src/textures/checkerboard.h:47-86): the value is
 * (1 - a) * spec1 + a * spec2 with a = 0 / 1 inside a check and the box-filtered area fraction across an edge. */
typedef enum mi_tex_type { MI_TEX_IMAGEMAP = 0, MI_TEX_CHECKERBOARD = 1 } mi_tex_type;

/* ---- lights */
typedef enum mi_light_type {
    MI_LIGHT_DIFFUSE_AREA = 0, /* src/lights/diffuse.cpp */
    MI_LIGHT_POINT,            /* src/lights/point.cpp */
    MI_LIGHT_DISTANT,          /* src/lights/distant.cpp */
    MI_LIGHT_INFINITE,         /* src/lights/infinite.cpp (environment light; mi_envmap) */
    MI_LIGHT_SPOT              /* src/lights/spot.cpp: pos, L = I, w2l, cos_total_width, cos_falloff_start */
} mi_light_type;

typedef struct mi_light {
    int32_t type;
    int32_t shape;     /* area: >=0 triangle index, <0 ~sphere index */
    int32_t two_sided;
    float area;        /* Shape::Area() */
    float L[MI_NSPEC]; /* Lemit / I / L; infinite: Spectrum(Lmap->Lookup((.5,.5), .5), Illuminant), i.e. Power() / (pi r^2) */
    float pos[3];      /* point: pLight */
    float dir[3];      /* distant: wLight (normalised, world) */
    float world_radius;
    float world_center[3];
    int32_t envmap;    /* infinite: index into mi_scene_desc.envmaps */
    float l2w[9], w2l[9]; /* infinite, spot: rotation part of LightToWorld / WorldToLight, row major */
    float cos_total_width, cos_falloff_start; /* spot */
} mi_light;

/* InfiniteAreaLight::Lmap (level 0 of the MIPMap<RGBSpectrum>, after the reference's power-of-two resampling)
 * and its sampling distribution (src/lights/infinite.cpp:43-83, src/core/sampling.h:123-147). The light looks
 * the map up with width 0 (bilinear on level 0, mipmap.h:271-281) and converts the RGB value to a spectrum per
 * lookup (Spectrum(rgb, SpectrumType::Illuminant), spectrum.cpp:98-180 with rgb_illum below). */
typedef struct mi_envmap {
    int32_t width, height;     /* Lmap->Width(), Height() */
    const float *rgb;          /* [height * width * 3] */
    int32_t nu, nv;            /* Distribution2D: nu = 2 * width, nv = 2 * height */
    const float *cond_func;    /* [nv * nu]       pConditionalV[v]->func */
    const float *cond_cdf;     /* [nv * (nu + 1)] pConditionalV[v]->cdf */
    const float *cond_func_int;/* [nv]            pConditionalV[v]->funcInt */
    const float *marg_func;    /* [nv]            pMarginal->func (= cond_func_int) */
    const float *marg_cdf;     /* [nv + 1] */
    float marg_func_int;
} mi_envmap;

/* Light-selection pmf (src/core/lightdistrib.cpp). For UNIFORM / POWER the host
 * supplies one Distribution1D (func[n_lights], cdf[n_lights+1], func_int[1]). For
 * SPATIAL only the voxel resolution is given (func/cdf/func_int NULL,
 * n_distributions 0): mi_pt_create estimates one pmf per voxel on the device with
 * the reference's 128-point Halton estimator (lightdistrib.cpp:232-300), because the
 * estimator needs Light::Sample_Li, which lives on the render side of this ABI. */
typedef enum mi_lightdistrib_type { MI_LD_UNIFORM = 0, MI_LD_POWER, MI_LD_SPATIAL } mi_lightdistrib_type;
typedef struct mi_lightdistrib {
    int32_t type;
    int32_t n_voxels[3]; /* SPATIAL only */
    uint32_t n_distributions;
    const float *func;
    const float *cdf;
    const float *func_int;
} mi_lightdistrib;

/* ---- camera (perspective only, src/cameras/perspective.cpp:45-146) */
typedef struct mi_camera {
    float raster_to_camera[16];
    float camera_to_world[16];
    float lens_radius, focal_distance;
    float shutter_open, shutter_close;
} mi_camera;

/* ---- film + reconstruction filter (src/core/film.cpp:50-112, film.h:123-163) */
typedef struct mi_film {
    int32_t full_res[2];
    int32_t cropped_bounds[4]; /* x0,y0,x1,y1 (max exclusive) */
    int32_t sample_bounds[4];  /* Film::GetSampleBounds() */
    float filter_radius[2];
    float filter_table[256];   /* 16x16, Film ctor */
    float scale;
    float max_sample_luminance;
} mi_film;

/* ---- samplers. HALTON (src/samplers/halton.cpp:65-127) and SOBOL (src/samplers/sobol.cpp, lowdiscrepancy.h:229-274) are
 * global samplers: sample (pixel, n) is a pure function of its index, so a parallel render reproduces the reference sample
 * for sample. RANDOM (src/samplers/random.cpp:42-60) draws from one PCG32 stream per film tile in the reference, in the order
 * its thread happens to consume them -- the count per sample depends on the path -- which no parallel schedule can
 * reproduce; here every camera sample owns a PCG32 stream of its own (sequence number = n * pixel count of the sample bounds +
 * pixel index, rng.h:98-105: distinct for every pixel and sample number, also when a pass renders sample numbers beyond
 * samples_per_pixel), consumed in the reference's order: the same estimator on other random numbers. */
/* ZEROTWO ("02sequence" / "lowdiscrepancy", src/samplers/zerotwosequence.cpp:53-69) and STRATIFIED (src/samplers/stratified.cpp:43-71)
 * are PixelSamplers (src/core/sampler.cpp:100-135): at StartPixel they tabulate, for each of `pixel_dims` sampled dimensions, one
 * 1D and one 2D value per sample of the pixel -- randomly scrambled / jittered and shuffled with the sampler's RNG -- and a
 * sample's Get1D / Get2D calls read the next 1D / 2D table until the tables run out, then fall back to that RNG. In the
 * reference the RNG is one stream per film tile, carried through its pixels and samples in the order the thread consumes it
 * (a count that depends on the paths). The convention here, identical on device and oracle, as for RANDOM: the tables of pixel
 * p come from a PCG32 stream of their own (sequence number = pixel index of the sample bounds), generated in the reference's
 * order (all 1D tables, then all 2D tables); the fall-back draws of sample n of pixel p come from the stream
 * (n + 1) * pixel count + pixel index. Same estimator, same stratification, other random numbers. */
typedef enum mi_sampler_type { MI_SAMPLER_HALTON = 0, MI_SAMPLER_SOBOL = 1, MI_SAMPLER_RANDOM = 2, MI_SAMPLER_ZEROTWO = 3, MI_SAMPLER_STRATIFIED = 4 } mi_sampler_type;
#define MI_SOBOL_MATRIX_SIZE 52 /* SobolMatrixSize, src/core/sobolmatrices.h:48 */
typedef struct mi_sampler {
    int64_t samples_per_pixel; /* SOBOL: rounded up to a power of two (sobol.h:52) */
    int32_t base_scales[2], base_exponents[2];
    int32_t sample_stride;
    int32_t mult_inverse[2];
    int32_t sample_at_pixel_center;
    int32_t n_dims;            /* dimensions with tables below */
    const int32_t *primes;     /* [n_dims] (also the bases of the spatial light distribution's probes, whatever the sampler) */
    const int32_t *prime_sums; /* [n_dims] offset of each base's permutation */
    const uint16_t *perms;     /* [n_perms] ComputeRadicalInversePermutations prefix */
    uint32_t n_perms;
    /* ABI v6 */
    int32_t type;              /* mi_sampler_type */
    int32_t sobol_resolution, sobol_log2_resolution; /* RoundUpPow2(max extent of the sample bounds), its log2 */
    int32_t n_sobol_dims;
    const uint32_t *sobol_matrices; /* [n_sobol_dims * 52] SobolMatrices32 */
    const uint64_t *sobol_vdc;      /* [52] VdCSobolMatrices[log2_resolution - 1] */
    const uint64_t *sobol_vdc_inv;  /* [52] VdCSobolMatricesInv[log2_resolution - 1] */
    /* ABI v9: the pixel samplers */
    int32_t pixel_dims;             /* ZEROTWO / STRATIFIED: nSampledDimensions ("integer dimensions", default 4) */
    int32_t x_samples, y_samples;   /* STRATIFIED: samples_per_pixel = x_samples * y_samples */
    int32_t jitter;                 /* STRATIFIED: "bool jitter" */
} mi_sampler;

typedef struct mi_integrator {
    int32_t max_depth;        /* CreatePathIntegrator, src/integrators/path.cpp:193 */
    float rr_threshold;
    int32_t pixel_bounds[4];  /* x0,y0,x1,y1 */
    int32_t n_ca_bands;       /* 1: Integrator "path". >1: Integrator "spectralpath" numCABands -- that many
                               * paths per camera sample on consecutive sampler dimensions, band s supplying
                               * spectrum bins [round(31/n)*s, min(round(31/n)*(s+1), 31))
                               * (src/integrators/spectralpath.cpp:258-318, 366) */
} mi_integrator;

typedef struct mi_scene_desc {
    uint32_t abi_version; /* MI_ABI_VERSION */
    uint32_t n_nodes;  const mi_bvh_node *nodes;
    uint32_t n_prims;  const mi_prim *prims;
    uint32_t n_tris;   const int32_t *tri_indices; /* 3 per triangle, global vertex ids */
                       const uint32_t *tri_mesh;   /* mesh id per triangle */
    uint32_t n_verts;  const float *P;  /* 3 per vertex, world space */
                       const float *N;  /* 3 per vertex, world space (0 when mesh has none) */
                       const float *UV; /* 2 per vertex */
    uint32_t n_meshes; const mi_mesh *meshes;
    uint32_t n_spheres; const mi_sphere *spheres;
    uint32_t n_materials; const mi_material *materials;
    uint32_t n_lights; const mi_light *lights;
    mi_lightdistrib light_distrib;
    mi_camera camera;
    mi_film film;
    mi_sampler sampler;
    mi_integrator integrator;
    float cie_y[MI_NSPEC]; /* SampledSpectrum::Y, for y() guards */
    uint32_t n_envmaps; const mi_envmap *envmaps;
    float rgb_illum[7][MI_NSPEC]; /* rgbIllum2Spect{White,Cyan,Magenta,Yellow,Red,Green,Blue} (spectrum.h:322-398) */
    uint32_t n_textures; const mi_texture *textures;   /* ABI v5 */
    uint32_t n_mipmaps;  const mi_mipmap *mipmaps;
    uint32_t n_instances; const mi_instance *instances; /* ABI v7 */
} mi_scene_desc;

/* Counters with the reference's STAT names (src/core/integrator.cpp:48,
 * src/core/scene.cpp:40-42, src/integrators/path.cpp:45-46). */
typedef struct mi_counters {
    uint64_t camera_rays;
    uint64_t regular_rays; /* Scene::Intersect calls */
    uint64_t shadow_rays;  /* Scene::IntersectP calls */
    uint64_t total_paths;  /* direct-lighting estimates */
    uint64_t zero_radiance_paths;
    uint64_t path_length_sum;
    uint64_t bvh_nodes_visited; /* build's own traversal, for B_ray */
    uint64_t tri_tests;
    uint64_t bad_samples;  /* NaN / negative / inf guards, integrator.cpp:295-316 */
    uint64_t iterations;   /* wavefront iterations of the last render */
    uint64_t extend_rays, extend_nodes, extend_tri_tests; /* closest-hit (extend) kernel only */
    uint64_t launches[3];  /* kernel launches of the last render: extend, shade, shadow */
} mi_counters;

#define MI_RENDER_FILM_ON_DEVICE 1u /* film_sum / weight_sum are device pointers */
#define MI_RENDER_ACCUMULATE 2u     /* do not clear the device film first */
typedef struct mi_render_params {
    int32_t shard_index, shard_count; /* 16x16 tiles with (tile_id % shard_count == shard_index) */
    uint32_t flags;
    uint32_t path_pool;   /* resident path slots; 0 = default */
    int64_t spp_override; /* 0 = use sampler.samples_per_pixel */
    int64_t sample_begin; /* first Halton sample number of this pass (0 = from the start);
                             a pass renders sample numbers [sample_begin, sample_begin + spp) */
    void *stream;         /* hipStream_t or NULL */
} mi_render_params;

typedef struct mi_pt mi_pt;

/* Create a renderer on HIP device `device_ordinal` and upload the scene. */
int mi_pt_create(const mi_scene_desc *scene, int device_ordinal, mi_pt **out);
/* Render. film_sum: [H*W*31] floats, pixel-major (pixel p, bin c at p*31+c) over
 * the cropped pixel bounds = Film::Pixel::L (sum of L*w*filterWeight, NOT
 * normalised, src/core/film.cpp:124-142); weight_sum: [H*W] = filterWeightSum.
 * Either may be NULL. Blocks until done. */
int mi_pt_render(mi_pt *pt, const mi_render_params *params, float *film_sum,
                 float *weight_sum, mi_counters *counters);
/* Device pointer of the resident film (layout [H*W][32]: 31 bins + weight) so a
 * collective can reduce in place; element count returned through n_floats. */
int mi_pt_device_film(mi_pt *pt, void **dev_ptr, uint64_t *n_floats);
/* Seconds of the last mi_pt_render: [0] = wall time of the whole render loop; then, from
 * HIP events recorded on the stream each launch goes to, the sum over launches per kernel
 * class: [1]=generate, [2]=extend (closest-hit traversal + its resolve), [3]=shade,
 * [4]=shadow, [5]=mis, [6]=closest-hit traversal kernel alone. The sub-renderers of one
 * render run concurrently, so [1..6] add up to more than [0]. */
int mi_pt_last_timings(mi_pt *pt, double *seconds, int n);
/* The path pool the last render ran on (summed over the renderer's sub-renderers): resident path slots and the device bytes
 * behind them. mi_render_params.path_pool = 0 asks for the default (one slot per camera sample of the shard, 4M .. 96M), which
 * takes at most 65 % of the free device memory and falls back to half, a quarter, ... if the allocation fails: this says what
 * was had. */
int mi_pt_pool_info(mi_pt *pt, uint64_t *slots, uint64_t *bytes);
void mi_pt_destroy(mi_pt *pt);
const char *mi_pt_last_error(void);

/* ---- Traversal-only entry point (parity of the BVH2 kernel on
 * recorded rays). rays: n x {o[3], d[3], tMax} floats (7 per ray);
 * hits: n x {prim (int32 as float bits), t, b0, b1} ; any_hit!=0 -> IntersectP
 * semantics (prim = 0/-1 only). Host pointers. */
int mi_pt_trace(mi_pt *pt, const float *rays, uint32_t n, int any_hit, float *hits);

/* The same question answered by the kernels a render launches (mi_pt_trace runs a plain per-ray routine, k_trace): ray i is
 * loaded into slot i of a path pool and a work list as the pipeline leaves it, traversed by the persistent wavefront kernel
 * of its class (batched state machine, cooperative leaf test, postponed quadrics, instance return entries) and committed by
 * that class's resolve step. Restates BVHAccel::Intersect / IntersectP (src/accelerators/bvh.cpp:662-738).
 *   mode 0: path rays -- k_trav<0>, k_resolve_extend, k_resolve_overflow; closest hit, tMax as given.
 *   mode 1: NEE shadow rays -- k_trav<1>, k_resolve_shadow, k_resolve_overflow; any hit (prim = 0 / -1); d is the
 *           unnormalised vector to the light sample and tMax must be 1 - 0.0001f (Interaction::SpawnRayTo).
 *   mode 2: BSDF-sampled MIS rays -- k_trav<2>, then the quadric step of k_resolve_mis (the same device function; that
 *           kernel consumes the hit in place); closest hit, tMax must be +infinity (Interaction::SpawnRay). (A render of a
 *           scene without instances (and without an alpha mask on an emitter's own mesh) asks these rays as visibility queries bounded by the sampled emitter,
 *           k_trav<3>, and falls back to this closest-hit form for the rays that does not settle; this call always runs
 *           the closest-hit kernel.)
 *   mode 3: the same rays as the visibility queries a render asks (k_trav<3>; only for scenes without instances and
 *           without an alpha mask on an emitter's own mesh): tMax is the end of the span in which the sampled emitter could be hit, the span starts at
 *           tMax (1 - 2^-8), no primitive is left out. hits[4i]: a primitive accepted in front of the span (any one: the
 *           kernel stops at the first), -1 if nothing was accepted up to tMax, -2 if something was accepted only inside the
 *           span (a render traces such a ray again in the reference's order); postponed quadrics are reported, not tested.
 * hits: as mi_pt_trace. extra (may be NULL): n x 4 words {b2 (float), instance of the hit (int32 bits, -1 = none),
 * the count of postponed quadrics as the traversal kernel left it (int32 bits: count | 0x100 on overflow),
 * the hit primitive as the traversal kernel left it, before the quadric step (int32 bits; -2 = the ray was never answered;
 * mode 1, whose kernel keeps one answer word per queue entry: 0 = occluded, -1 = not)}. Host pointers; n <= 2^24. */
int mi_pt_trace_wavefront(mi_pt *pt, const float *rays, uint32_t n, int mode, float *hits, float *extra);

/* Parity tool for the scalar helpers under the shape and sampling code, each run on the device for n inputs (x, y: 2 floats per
 * element; out: 3 floats per element) -- the reference's own tests of them are restated over this entry point and the oracle's:
 *   op 0: NextFloatUp(x0), NextFloatDown(x0)                     pbrt.h:244-268  (OffsetRayOrigin; tests/fp_tests.cpp:29-47)
 *   op 1..4: EFloat(x0, err x1) {+, -, *, /} EFloat(y0, err y1)  efloat.h:48-200 (Sphere::Intersect; tests/fp_tests.cpp:166-260)
 *            -> value, lower bound, upper bound
 *   op 5: FindInterval over the array 0, 1, ..., 9 with the predicate a[i] <= x0, as Distribution1D::SampleDiscrete runs it
 *         (pbrt.h:405-418, sampling.h:91-98; tests/find_interval.cpp:8) -> the interval
 * Host pointers; needs no scene. */
int mi_pt_math_probe(int device_ordinal, int op, uint32_t n, const float *x, const float *y, float *out);

/* Parity tool for image textures: MIPMap<RGBSpectrum>::Lookup(st, dstdx, dstdy) (src/core/mipmap.h:281-319) of texture
 * `tex` for n queries on the device. queries: 6 floats each (s, t, dsdx, dtdx, dsdy, dtdy) in texture space, i.e. after
 * the UVMapping2D; rgb: 3 floats per query. */
int mi_pt_texture_lookup(mi_pt *pt, int32_t tex, uint32_t n, const float *queries, float *rgb);

/* Parity tool for the spatial light-selection strategy: the per-voxel Distribution1D tables mi_pt_create estimated on
 * the device (SpatialLightDistribution::ComputeDistribution, src/core/lightdistrib.cpp:232-300; the reference fills its
 * hash table lazily, this build tabulates every voxel). Voxel (x, y, z) has index (z * n_voxels[1] + y) * n_voxels[0] + x.
 * func: [n_voxels * n_lights] floats (Distribution1D::func of voxel v at v * n_lights), func_int: [n_voxels]
 * (Distribution1D::funcInt); either may be NULL. capacity_voxels must be >= the voxel count. MI_ERR_INVALID when the
 * scene does not use "lightsamplestrategy" "spatial" (or has no lights). Host pointers. */
int mi_pt_light_distribution(mi_pt *pt, float *func, float *func_int, uint64_t capacity_voxels);

/* ---- HLBVH build on the device (Accelerator "bvh" "string splitmethod" "hlbvh"; BVHAccel::HLBVHBuild,
 * src/accelerators/bvh.cpp:404-638): Morton codes, stable radix sort, one LBVH treelet per run of equal top 12 bits
 * (emitLBVH), and the treelets flattened into the depth-first 32-byte node array. The SAH tree over the (at most 4096)
 * treelet roots (buildUpperSAH, bvh.cpp:534-638) is small serial host work: the caller supplies it as `upper`, which
 * receives the treelets' root bounds (6 floats each: min xyz, max xyz) and node counts and returns the upper interior
 * nodes (children given as final indices), their own final indices, the total node count and the final index of each
 * treelet's first node. prim_bounds: n x {min xyz, max xyz}; nodes_out: capacity nodes_capacity (2 n is always enough);
 * ordered_out: n primitive numbers in leaf order. The tree is the one the reference builds on one thread (leaves in
 * Morton order): the host restatement in libmipt_host.so builds the same nodes, bit for bit. */
typedef int (*mi_bvh_upper_fn)(void *user, uint32_t n_treelets, const float *root_bounds, const int32_t *treelet_sizes,
                               mi_bvh_node *upper_nodes, int32_t *upper_index, uint32_t *n_upper, uint32_t *n_total,
                               int32_t *treelet_offset);
int mi_bvh_build_hlbvh(int device_ordinal, const float *prim_bounds, uint32_t n, int32_t max_prims_in_node, mi_bvh_upper_fn upper,
                       void *user, mi_bvh_node *nodes_out, uint32_t nodes_capacity, uint32_t *n_nodes, int32_t *ordered_out,
                       double *seconds);
const char *mi_bvh_last_error(void);

/* Parity tool: the state of ONE camera sample (pixel px, py; Halton sample number `sample`) vertex by vertex, for a
 * side-by-side comparison with the oracle's log of the same sample (oracle_path_log) when a film differs. One record of
 * MI_PATH_RECORD_FLOATS floats per path vertex (one wavefront iteration):
 *   [0] bounces  [1] hit primitive (-1: escaped)  [2] sampler dimension before the vertex  [3] 1 if the path ended here
 *   [4..6] incoming ray origin  [7] t of the hit  [8..10] incoming ray direction  [11] etaScale before
 *   [12..14] next ray origin  [15] sampler dimension after  [16..18] next ray direction  [19] etaScale after
 *   [20..50] beta after the vertex  [51..81] L after the vertex's direct lighting has been added
 * The device film is cleared by this call. */
#define MI_PATH_RECORD_FLOATS 96
int mi_pt_debug_path(mi_pt *pt, int32_t px, int32_t py, int64_t sample, int32_t max_records, float *records, int32_t *n_records);

#ifdef __cplusplus
}
#endif
#endif /* MI_PT_H */
