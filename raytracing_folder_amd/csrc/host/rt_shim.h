// rt_shim.h -- the reference's driver surface on top of the C ABI.
//
// The reference's UI calls three free functions -- BeginRender() (must return immediately, the
// render proceeds in the background), StopRender() and saveImage() (FIN/viewport.cpp:35-37,
// 443-453) -- and polls renderImage.IsRenderDone() / reads renderImage.GetPixels() while the
// render runs (FIN/viewport.cpp:367,390-409).  rt::Renderer keeps exactly that contract:
// same method names and meaning, same RenderImage container (FIN/include/scene.h:540-656),
// but the pixels come from the HIP kernels through rt_render_begin().
#ifndef RT_HOST_SHIM_H
#define RT_HOST_SHIM_H

#include <string>
#include <vector>

#include "rt_scene.h"

namespace rt {

class RenderImage {
    std::vector<uint8_t> img;          // Color24[width*height]
    std::vector<float> zbuffer;
    std::vector<uint8_t> zbufferImg, sampleCount, sampleCountImg;
    int width = 0, height = 0;
    std::vector<rt_job *> jobs;        // progress sources while a render is live (one job per device)
    int finalPixels = 0;
public:
    void Init(int w, int h);
    int GetWidth() const { return width; }
    int GetHeight() const { return height; }
    uint8_t *GetPixels() { return img.data(); }
    float *GetZBuffer() { return zbuffer.data(); }
    uint8_t *GetZBufferImage() { return zbufferImg.data(); }
    uint8_t *GetSampleCount() { return sampleCount.data(); }
    uint8_t *GetSampleCountImage() { return sampleCountImg.data(); }
    int GetNumRenderedPixels() const;
    bool IsRenderDone() const { return GetNumRenderedPixels() >= width * height; }
    void ComputeZBufferImage();        // scene.h:591-613
    int ComputeSampleCountImage();     // scene.h:615-637
    bool SaveImage(const char *filename) const { return WritePNG(filename, img.data(), width, height, 3); }
    bool SaveZImage(const char *filename) const { return WritePNG(filename, zbufferImg.data(), width, height, 1); }
    bool SaveSampleCountImage(const char *filename) const { return WritePNG(filename, sampleCountImg.data(), width, height, 1); }
    void AttachJob(rt_job *j) { jobs.push_back(j); }
    void DetachJobs(int pixels) { jobs.clear(); finalPixels = pixels; }
};

class Renderer {
public:
    Scene scene;                 // rootNode, camera, materials, lights, objList, environment, background
    RenderImage renderImage;
    rt_params params;            // the reference's #defines as run-time values (incl. MAX_NUM_OF_PHOTON / PHOTON_BOUNCE)
    // devices to render on: empty (the default) = every gfx950 device of the node (rt_device_count()).  The reference spreads
    // the pixels over 2 x hardware_concurrency threads that share one counter (FIN/main.cpp:71-78, 987-997); here device r of
    // N takes the interleaved 32 x 8 tiles r, r+N, ... and writes them into the same RenderImage.
    std::vector<int> devices;
    std::string photonDump;      // where generatePhotonMap's .dat goes (FIN/main.cpp:398 hard-codes a path); empty = no dump

    Renderer();
    ~Renderer();
    int LoadScene(const char *filename);          // FIN/xmlload.cpp:65-132: 1 on success, 0 on failure
    // balanced photon array (index 0 unused), the product of generatePhotonMap (FIN/main.cpp:350-402)
    bool SetPhotonMap(const rt_photon *balanced, uint32_t n_stored);
    // Like the reference's (FIN/main.cpp:984-998): generatePhotonMap first -- when no map was set with SetPhotonMap and the
    // FIN model renders -- then the workers; but BOTH on background threads, so the call returns immediately (progress stays
    // 0 while the photon pass runs).  false + LastError() when the GPU path cannot start.
    bool BeginRender();
    void StopRender();
    bool WaitRender();           // joins the background job (the reference polls IsRenderDone instead)
    void saveImage(const char *image = "prj13box.png", const char *samples = "prj13box_sc.png", const char *zimage = nullptr);
    const std::string &LastError() const { return error; }
    const rt_stats &Stats() const { return stats; }         // summed over the devices
    const rt_setup_ms &PhotonPassMs() const { return setup; }   // stage times of the photon pass BeginRender ran (zero: none)
    int NumDevices() const { return (int)jobs.size(); }
private:
    rt_scene *handle = nullptr;
    std::vector<rt_job *> jobs;
    rt_setup_ms setup{};
    std::string error;
    rt_stats stats{};
    bool lowered = false;
};

}  // namespace rt
#endif
