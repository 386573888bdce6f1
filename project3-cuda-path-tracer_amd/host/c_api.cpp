// c_api.cpp -- plain-C view of the host-side Scene loader and PNG writer for the Python mirror
// (project3-cuda-path-tracer_amd/__init__.py) and for other-language hosts.
#include <cstring>
#include <stdexcept>

#include "image.h"
#include "scene.h"

extern "C" {

void *pth_scene_load(const char *path) {
    try {
        return new Scene(path, false);
    } catch (const std::exception &) {
        return NULL;
    }
}
void pth_scene_free(void *s) { delete static_cast<Scene *>(s); }
int pth_scene_num_geoms(void *s) { return (int)static_cast<Scene *>(s)->geoms.size(); }
int pth_scene_num_materials(void *s) { return (int)static_cast<Scene *>(s)->materials.size(); }
const void *pth_scene_geoms(void *s) { return static_cast<Scene *>(s)->geoms.data(); }
const void *pth_scene_materials(void *s) { return static_cast<Scene *>(s)->materials.data(); }
const void *pth_scene_camera(void *s) { return &static_cast<Scene *>(s)->state.camera; }
int pth_scene_iterations(void *s) { return (int)static_cast<Scene *>(s)->state.iterations; }
int pth_scene_depth(void *s) { return static_cast<Scene *>(s)->state.traceDepth; }
const char *pth_scene_image_name(void *s) { return static_cast<Scene *>(s)->state.imageName.c_str(); }
int pth_scene_num_meshes(void *s) { return (int)static_cast<Scene *>(s)->meshes.size(); }
int pth_scene_mesh_geom(void *s, int i) { return static_cast<Scene *>(s)->meshes[i].geom; }
int pth_scene_mesh_ntris(void *s, int i) { return (int)(static_cast<Scene *>(s)->meshes[i].tris.size() / 9); }
const float *pth_scene_mesh_tris(void *s, int i) { return static_cast<Scene *>(s)->meshes[i].tris.data(); }
// vertex normals (ntris x 9) / face materials (ntris) of mesh i, or NULL when the OBJ gave none
const float *pth_scene_mesh_normals(void *s, int i) {
    const Mesh &m = static_cast<Scene *>(s)->meshes[i];
    return m.normals.empty() ? NULL : m.normals.data();
}
const int *pth_scene_mesh_materials(void *s, int i) {
    const Mesh &m = static_cast<Scene *>(s)->meshes[i];
    return m.mats.empty() ? NULL : m.mats.data();
}
void pth_scene_set_resolution(void *s, int w, int h) { static_cast<Scene *>(s)->setResolution(w, h); }

// saveImage (reference src/main.cpp:49-70) on a W*H*3 running sum: /samples, X mirror, PNG
int pth_save_png(const char *basename, const float *rgb_sum, int w, int h, float samples) {
    image img(w, h);
    for (int x = 0; x < w; x++)
        for (int y = 0; y < h; y++) {
            const float *p = rgb_sum + 3 * ((size_t)x + (size_t)y * w);
            img.setPixel(w - 1 - x, y, lin::vec3(p[0], p[1], p[2]) / samples);
        }
    return img.savePNG(basename) ? 0 : -1;
}

// same image as pth_save_png, written as Radiance .hdr (reference image::saveHDR, src/image.cpp:41-45)
int pth_save_hdr(const char *basename, const float *rgb_sum, int w, int h, float samples) {
    image img(w, h);
    for (int x = 0; x < w; x++)
        for (int y = 0; y < h; y++) {
            const float *p = rgb_sum + 3 * ((size_t)x + (size_t)y * w);
            img.setPixel(w - 1 - x, y, lin::vec3(p[0], p[1], p[2]) / samples);
        }
    return img.saveHDR(basename) ? 0 : -1;
}

}  // extern "C"
