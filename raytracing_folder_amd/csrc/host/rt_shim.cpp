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

// RenderImage::ComputeZBufferImage / ComputeSampleCountImage, FIN/include/scene.h:591-637 (rt_image.cpp)
void RenderImage::ComputeZBufferImage()
{
    zbufferImg.assign((size_t)width * height, 0);
    ZBufferImage(zbuffer.data(), zbufferImg.size(), zbufferImg.data());
}

int RenderImage::ComputeSampleCountImage()
{
    sampleCountImg.assign((size_t)width * height, 0);
    return SampleCountImage(sampleCount.data(), sampleCountImg.size(), sampleCountImg.data());
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
