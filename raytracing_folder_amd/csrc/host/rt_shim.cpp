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

int RenderImage::GetNumRenderedPixels() const
{
    if (jobs.empty()) return finalPixels;
    int n = 0;
    for (rt_job *j : jobs) n += rt_render_progress(j);
    return n;
}

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
    for (rt_job *j : jobs) { rt_render_stop(j); rt_render_wait(j); }
    renderImage.DetachJobs(0);
    for (rt_job *j : jobs) rt_job_destroy(j);
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
    if (!jobs.empty()) { error = "a render is already running"; return false; }
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
    if (st == RT_OK) st = rt_scene_set_photon_dump(handle, photonDump.empty() ? nullptr : photonDump.c_str());
    if (st != RT_OK) { error = rt_last_error(); return false; }
    // one job per device: device r of N renders the interleaved tiles r, r+N, ... into the SAME caller-owned buffers (a job
    // writes only the pixels of its own tiles); whichever job comes first runs the photon pass, the others wait for it
    std::vector<int> devs = devices;
    if (devs.empty()) for (int i = 0, n = rt_device_count(); i < n; i++) devs.push_back(i);
    if (devs.empty()) devs.push_back(0);              // no gfx950 device: rt_render_begin reports it (there is no CPU path)
    const int N = (int)devs.size();
    for (int r = 0; r < N; r++) {
        const rt_tile_range mine = {32, 8, r, N};
        rt_job *job = nullptr;
        st = rt_render_begin(handle, &d.camera, &params, &mine, devs[r], renderImage.GetPixels(), renderImage.GetZBuffer(),
                             renderImage.GetSampleCount(), &job);
        if (st != RT_OK) {
            error = rt_last_error();
            for (rt_job *j : jobs) { rt_render_stop(j); rt_render_wait(j); }
            renderImage.DetachJobs(0);
            for (rt_job *j : jobs) rt_job_destroy(j);
            jobs.clear();
            return false;
        }
        jobs.push_back(job);
        renderImage.AttachJob(job);
    }
    return true;
}

void Renderer::StopRender() { for (rt_job *j : jobs) rt_render_stop(j); }

bool Renderer::WaitRender()
{
    if (jobs.empty()) return true;
    bool ok = true;
    int pixels = 0;
    memset(&stats, 0, sizeof stats);
    memset(&setup, 0, sizeof setup);
    for (rt_job *j : jobs) {
        const rt_status st = rt_render_wait(j);
        if (st != RT_OK) { error = rt_last_error(); ok = false; }
        rt_stats one;
        if (rt_job_stats(j, &one) == RT_OK) {
            stats.rays_primary += one.rays_primary; stats.rays_shadow += one.rays_shadow; stats.rays_reflect += one.rays_reflect;
            stats.rays_refract += one.rays_refract; stats.photon_queries += one.photon_queries; stats.photons_visited += one.photons_visited;
            stats.pixels += one.pixels; stats.samples += one.samples;
            if (one.ms_total > stats.ms_total) stats.ms_total = one.ms_total;
        }
        rt_setup_ms su;
        if (rt_job_setup_ms(j, &su) == RT_OK && su.total > 0) setup = su;
        pixels += rt_render_progress(j);
    }
    renderImage.DetachJobs(pixels);
    for (rt_job *j : jobs) rt_job_destroy(j);
    jobs.clear();
    return ok;
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
