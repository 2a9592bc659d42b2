/*
 * capsaicin_scene.h — C ABI of the host-side scene surface kept from the reference: OBJ/MTL ingestion into the
 * pooled GeometryStorage layout.  Replaces AssetLoadSystem::LoadObjFile + the CPU half of CreateGeometryStorage
 * (src/systems/asset_load_system.cpp:43-160, 162-233) and the tinyobjloader call it wraps (:54-55; the
 * submodule is empty in the reference tree, its consumed behaviour is re-implemented).
 * No GPU is touched by these functions.
 */
#ifndef CAPSAICIN_SCENE_H
#define CAPSAICIN_SCENE_H

#include <stdint.h>

#include "capsaicin_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct CapGeometry CapGeometry;

/* LoadObjFile (asset_load_system.cpp:43-160).  mtl_dir: directory searched for `mtllib` files ("" or NULL =
 * directory of the OBJ; the reference passes "../../../assets/", :55).  A missing MTL file is a warning, not an
 * error, and leaves every mesh untextured.  Parse errors return CAP_ERR_IO (the reference throws). */
int  cap_obj_load(const char* obj_path, const char* mtl_dir, CapGeometry** out_geometry);
void cap_geometry_free(CapGeometry* g);
/* Threads cap_obj_load parses the hot records of a large file on (process-wide; 0 = default: one per hardware thread, at most 8, for
 * files above 1 MB; 1 = the sequential parser only).  The result does not depend on it: tests/test_obj_loader.py. */
void cap_obj_set_threads(int threads);

typedef struct CapGeometryView
{
    const float*       positions; /* 3 * vertex_count */
    const float*       normals;   /* 3 * vertex_count */
    const float*       texcoords; /* 2 * vertex_count */
    const uint32_t*    indices;   /* index_count, mesh-local */
    const CapMeshDesc* meshes;    /* mesh_count */
    uint32_t           vertex_count;
    uint32_t           index_count;
    uint32_t           mesh_count;
    uint32_t           texture_count;  /* distinct diffuse_texname values, in first-use order */
    uint32_t           material_count; /* materials parsed from the MTL files that could be opened */
} CapGeometryView;

int         cap_geometry_view(const CapGeometry* g, CapGeometryView* out);
const char* cap_geometry_texture_name(const CapGeometry* g, uint32_t texture_index);
const char* cap_geometry_warning(const CapGeometry* g);
/* EXT: one CapMaterial per mesh from the MTL's Kd/Ks/Ns/Ke of the mesh's first face material (mesh_count entries).
 * Meshes without a material get kd = 0.75^2.2 (the reference's untextured albedo), ks = ke = 0. */
int cap_geometry_materials(const CapGeometry* g, CapMaterial* out_materials);

/* Convenience: cap_scene_upload(ctx, view...) */
int cap_scene_upload_geometry(CapContext* ctx, const CapGeometry* g);

/* Texture file -> what TextureSystem hands to the GPU (texture_system.cpp:41-45: stbi_load(file, &w, &h, &n, 4)): 8-bit RGBA,
 * rows top to bottom, grey replicated, alpha 255 when the file has none.  Decodes JPEG (Huffman-coded baseline / progressive),
 * PNG, BMP, TGA and binary PNM (capsaicin_amd/csrc/image_decode.cpp, jpeg_decode.cpp), pixel for pixel what stbi_load returns for the
 * same bytes; the container is recognised from the bytes in stb's order (TGA, which has no signature, last), name_hint (may be
 * NULL) is not consulted.  Anything else, and any file that is damaged or shorter than its header demands, returns
 * CAP_ERR_UNSUPPORTED -- the caller then does what the reference does for a missing file: a warning and
 * a 1x1 black texel (texture_system.cpp:50-56).  Release the pixels with cap_image_free. */
int  cap_image_decode(const uint8_t* bytes, size_t size, const char* name_hint, uint8_t** out_rgba8, uint32_t* out_width,
                      uint32_t* out_height);
void cap_image_free(uint8_t* rgba8);

/* Host-side tree build used by cap_bvh_build in SAH mode, callable without a GPU (tools, tests): triangle boxes in, the
 * device node layout out.  tri_boxes: n x 8 floats (lo.xyz, -, hi.xyz, -); nodes: 16 floats per internal node, n - 1 of
 * them (box of child 0, box of child 1, child0, child1, traversal child0, traversal child1 as int bits; a child < 0 is
 * ~leaf position); order: n triangle ids in leaf order; depth: internal nodes on the longest root-to-leaf path. */
int cap_host_sah_build(const float* tri_boxes, uint32_t n, float* nodes, uint32_t* order, uint32_t* depth);

/* Host-side collapse of that binary tree into the compressed 8-wide view the extension- and shadow-ray kernels walk
 * (capsaicin_amd/csrc/cap_wide.h; replaces, with the build above, the driver's acceleration-structure build behind
 * blas_system.cpp:65 / tlas_system.cpp:72), callable without a GPU.  nodes: the n - 1 binary nodes as above (NULL when n < 2);
 * scene_lo / scene_hi: bounds of all triangles.  wide_nodes: room for wide_capacity nodes of 20 uint32 each (n nodes always
 * suffice); tri_src: n entries, wide-order triangle record i = leaf position tri_src[i]; info: {node count, depth, leading
 * nodes that form the top levels}.  CAP_ERR_INVALID_ARG when the capacity is too small. */
int cap_host_wide_build(const float* nodes, uint32_t n, const float* scene_lo, const float* scene_hi, uint32_t* wide_nodes,
                        uint32_t wide_capacity, uint32_t* tri_src, uint32_t* info);

#ifdef __cplusplus
}
#endif
#endif
