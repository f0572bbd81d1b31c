// rt_shim.cpp -- BeginRender/StopRender/saveImage + RenderImage over the C ABI (see rt_shim.h).
#include "rt_shim.h"

#include <cstring>

namespace rt {

void RenderImage::Init(int w, int h)
{
    width = w; height = h;
    img.assign((size_t)w * h * 3, 0);
    zbuffer.assign((size_t)w * h, 0.0f);
    sampleCount.assign((size_t)w * h, 0);
    zbufferImg.clear(); sampleCountImg.clear();
    finalPixels = 0;
}

int RenderImage::GetNumRenderedPixels() const { return job ? rt_render_progress(job) : finalPixels; }

// RenderImage::ComputeZBufferImage, FIN/include/scene.h:591-613
void RenderImage::ComputeZBufferImage()
{
    const size_t size = (size_t)width * height;
    zbufferImg.assign(size, 0);
    const float BIG = 1.0e30f;
    float zmin = BIG, zmax = 0;
    for (size_t i = 0; i < size; i++) {
        if (zbuffer[i] == BIG) continue;
        if (zmin > zbuffer[i]) zmin = zbuffer[i];
        if (zmax < zbuffer[i]) zmax = zbuffer[i];
    }
    for (size_t i = 0; i < size; i++) {
        if (zbuffer[i] == BIG) { zbufferImg[i] = 0; continue; }
        const float f = (zmax - zbuffer[i]) / (zmax - zmin);
        int c = int(f * 255);
        if (c < 0) c = 0;
        if (c > 255) c = 255;
        zbufferImg[i] = (uint8_t)c;
    }
}

// RenderImage::ComputeSampleCountImage, FIN/include/scene.h:615-637
int RenderImage::ComputeSampleCountImage()
{
    const size_t size = (size_t)width * height;
    sampleCountImg.assign(size, 0);
    uint8_t smin = 255, smax = 0;
    for (size_t i = 0; i < size; i++) { if (smin > sampleCount[i]) smin = sampleCount[i]; if (smax < sampleCount[i]) smax = sampleCount[i]; }
    if (smax != smin)
        for (size_t i = 0; i < size; i++) {
            int c = (255 * (sampleCount[i] - smin)) / (smax - smin);
            if (c < 0) c = 0;
            if (c > 255) c = 255;
            sampleCountImg[i] = (uint8_t)c;
        }
    return smax;
}

Renderer::Renderer() { rt_params_default(&params); rt_scene_create(&handle); }

Renderer::~Renderer()
{
    if (job) { rt_render_stop(job); rt_render_wait(job); renderImage.DetachJob(0); rt_job_destroy(job); }
    rt_scene_destroy(handle);
}

int Renderer::LoadScene(const char *filename)
{
    lowered = false;
    if (!rt::LoadScene(scene, filename, &error)) return 0;
    renderImage.Init(scene.camera.imgWidth, scene.camera.imgHeight);
    return 1;
}

bool Renderer::SetPhotonMap(const rt_photon *balanced, uint32_t n_stored)
{
    if (rt_scene_set_photons(handle, balanced, n_stored) != RT_OK) { error = rt_last_error(); return false; }
    return true;
}

bool Renderer::BeginRender()
{
    if (job) { error = "a render is already running"; return false; }
    SceneData d;
    if (!Lower(scene, d, &error)) return false;
    rt_status st = rt_scene_set_nodes(handle, d.nodes.data(), (int32_t)d.nodes.size());
    for (size_t m = 0; st == RT_OK && m < d.meshes.size(); m++) {
        const MeshData &md = d.meshes[m];
        st = rt_scene_set_mesh(handle, (int32_t)m, md.v.data(), (int32_t)(md.v.size() / 3), md.f.data(), (int32_t)(md.f.size() / 3),
                               md.vn.data(), (int32_t)(md.vn.size() / 3), md.fn.data(), md.nodes.data(), (int32_t)md.nodes.size(),
                               md.elements.data());
        if (st == RT_OK && !md.vt.empty())
            st = rt_scene_set_mesh_texcoords(handle, (int32_t)m, md.vt.data(), (int32_t)(md.vt.size() / 3), md.ft.data());
    }
    if (st == RT_OK) st = rt_scene_set_materials(handle, d.materials.data(), (int32_t)d.materials.size());
    if (st == RT_OK) st = rt_scene_set_lights(handle, d.lights.data(), (int32_t)d.lights.size());
    if (st == RT_OK) st = rt_scene_set_environment(handle, d.env, d.bg);
    if (st == RT_OK) st = rt_scene_set_textures(handle, d.textures.data(), (int32_t)d.textures.size(), d.texels.data(), (uint64_t)d.texels.size());
    if (st == RT_OK) st = rt_scene_set_material_maps(handle, d.material_maps.data(), d.material_maps.empty() ? 0 : (int32_t)d.materials.size());
    if (st == RT_OK) st = rt_scene_set_environment_maps(handle, &d.env_map, &d.bg_map);
    if (st != RT_OK) { error = rt_last_error(); return false; }
    if (renderImage.GetWidth() != d.camera.width || renderImage.GetHeight() != d.camera.height)
        renderImage.Init(d.camera.width, d.camera.height);
    const rt_tile_range all = {32, 8, 0, 1};
    st = rt_render_begin(handle, &d.camera, &params, &all, device, renderImage.GetPixels(), renderImage.GetZBuffer(),
                         renderImage.GetSampleCount(), &job);
    if (st != RT_OK) { error = rt_last_error(); job = nullptr; return false; }
    renderImage.AttachJob(job);
    return true;
}

void Renderer::StopRender() { if (job) rt_render_stop(job); }

bool Renderer::WaitRender()
{
    if (!job) return true;
    const rt_status st = rt_render_wait(job);
    if (st != RT_OK) error = rt_last_error();
    rt_job_stats(job, &stats);
    renderImage.DetachJob(rt_render_progress(job));
    rt_job_destroy(job);
    job = nullptr;
    return st == RT_OK;
}

// saveImage(), FIN/main.cpp:1000-1007
void Renderer::saveImage(const char *image, const char *samples, const char *zimage)
{
    renderImage.ComputeZBufferImage();
    if (zimage) renderImage.SaveZImage(zimage);
    if (image) renderImage.SaveImage(image);
    renderImage.ComputeSampleCountImage();
    if (samples) renderImage.SaveSampleCountImage(samples);
}

}  // namespace rt
