// rt_image.cpp -- image file reading for textures: what TextureFile::Load (FIN/texture.cpp:57-91;
// FIN = /root/reference/RayTracingFinal/RayTracingFinal) gets from the vendored lodepng
// (`lodepng::decode(d, w, h, name, LCT_RGB)`: any PNG colour type converted to 8-bit RGB) and from
// its own LoadPPM (:33-53).  Own implementation: zlib inflate (stored / fixed / dynamic Huffman),
// PNG scanline filters, non-interlaced images of every legal bit depth (1/2/4-bit grey and palette
// samples unpacked and scaled like lodepng does, 16 keeps the high byte), colour types grey, RGB,
// palette, grey+alpha, RGBA (alpha dropped, as LCT_RGB does).  Also the two derived images of
// RenderImage (scene.h:591-637).
#include "rt_scene.h"

#include <cctype>
#include <cstdio>
#include <cstdlib>

namespace rt {
namespace {

struct BitReader {
    const uint8_t *p; size_t n, pos = 0; uint32_t bitbuf = 0; int bits = 0; bool bad = false;
    uint32_t get(int k)
    {
        while (bits < k) {
            if (pos >= n) { bad = true; return 0; }
            bitbuf |= (uint32_t)p[pos++] << bits; bits += 8;
        }
        const uint32_t v = bitbuf & ((1u << k) - 1u);
        bitbuf >>= k; bits -= k;
        return k ? v : 0;
    }
    void align() { bitbuf = 0; bits = 0; }
};

struct Huffman {
    uint16_t count[16]; uint16_t symbol[320];
    void build(const uint8_t *len, int n)
    {
        for (int i = 0; i < 16; i++) count[i] = 0;
        for (int i = 0; i < n; i++) count[len[i]]++;
        count[0] = 0;
        uint16_t offs[16]; offs[1] = 0;
        for (int i = 1; i < 15; i++) offs[i + 1] = offs[i] + count[i];
        for (int i = 0; i < n; i++) if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
    }
    int decode(BitReader &br) const
    {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len <= 15; len++) {
            code |= (int)br.get(1);
            if (br.bad) return -1;
            const int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        return -1;
    }
};

bool inflate(const uint8_t *src, size_t n, std::vector<uint8_t> &out)
{
    static const uint16_t lbase[] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t lext[] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dext[] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    if (n < 2) return false;
    BitReader br{src + 2, n - 2};             // skip the zlib header (CMF, FLG)
    int last;
    do {
        last = (int)br.get(1);
        const int type = (int)br.get(2);
        if (br.bad) return false;
        if (type == 0) {
            br.align();
            if (br.pos + 4 > br.n) return false;
            const uint32_t len = br.p[br.pos] | (br.p[br.pos + 1] << 8);
            br.pos += 4;
            if (br.pos + len > br.n) return false;
            out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
            br.pos += len;
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            uint8_t lens[320];
            if (type == 1) {
                int i = 0;
                for (; i < 144; i++) lens[i] = 8;
                for (; i < 256; i++) lens[i] = 9;
                for (; i < 280; i++) lens[i] = 7;
                for (; i < 288; i++) lens[i] = 8;
                lit.build(lens, 288);
                for (i = 0; i < 30; i++) lens[i] = 5;
                dist.build(lens, 30);
            } else {
                const int nlen = (int)br.get(5) + 257, ndist = (int)br.get(5) + 1, ncode = (int)br.get(4) + 4;
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; i++) cl[order[i]] = (uint8_t)br.get(3);
                Huffman clh;
                clh.build(cl, 19);
                int i = 0;
                while (i < nlen + ndist) {
                    const int sym = clh.decode(br);
                    if (sym < 0) return false;
                    if (sym < 16) lens[i++] = (uint8_t)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) { if (i == 0) return false; val = lens[i - 1]; rep = 3 + (int)br.get(2); }
                        else if (sym == 17) rep = 3 + (int)br.get(3);
                        else rep = 11 + (int)br.get(7);
                        if (i + rep > nlen + ndist) return false;
                        while (rep--) lens[i++] = (uint8_t)val;
                    }
                }
                lit.build(lens, nlen);
                dist.build(lens + nlen, ndist);
            }
            for (;;) {
                const int sym = lit.decode(br);
                if (sym < 0 || br.bad) return false;
                if (sym < 256) out.push_back((uint8_t)sym);
                else if (sym == 256) break;
                else {
                    const int li = sym - 257;
                    if (li >= 29) return false;
                    const int len = lbase[li] + (int)br.get(lext[li]);
                    const int ds = dist.decode(br);
                    if (ds < 0 || ds >= 30) return false;
                    const size_t d = dbase[ds] + br.get(dext[ds]);
                    if (d > out.size()) return false;
                    const size_t start = out.size() - d;
                    for (int k = 0; k < len; k++) out.push_back(out[start + k]);
                }
            }
        } else return false;
    } while (!last);
    return true;
}

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

bool load_png(const std::vector<uint8_t> &f, int &w, int &h, std::vector<uint8_t> &rgb, std::string *err)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (f.size() < 8 || memcmp(f.data(), sig, 8) != 0) { if (err) *err = "not a PNG file"; return false; }
    size_t pos = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    while (pos + 12 <= f.size()) {
        const uint32_t len = be32(&f[pos]);
        const char *tag = (const char *)&f[pos + 4];
        if (pos + 12 + (size_t)len > f.size()) break;
        const uint8_t *d = &f[pos + 8];
        if (!memcmp(tag, "IHDR", 4) && len >= 13) { w = (int)be32(d); h = (int)be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12]; }
        else if (!memcmp(tag, "PLTE", 4)) plte.assign(d, d + len);
        else if (!memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
        else if (!memcmp(tag, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (w <= 0 || h <= 0 || interlace != 0) { if (err) *err = "unsupported PNG (need a non-interlaced image)"; return false; }
    // IHDR is untrusted input: bound the pixel count before any size arithmetic (2^28 pixels = the render limit)
    if ((unsigned long long)w * (unsigned long long)h > (1ull << 28)) { if (err) *err = "PNG too large (more than 2^28 pixels)"; return false; }
    int ch;
    switch (ctype) { case 0: ch = 1; break; case 2: ch = 3; break; case 3: ch = 1; break; case 4: ch = 2; break; case 6: ch = 4; break; default: if (err) *err = "bad PNG colour type"; return false; }
    // legal depths per colour type (PNG 1.2, table 11.1): grey 1/2/4/8/16, palette 1/2/4/8, the rest 8/16
    const bool sub_byte = depth == 1 || depth == 2 || depth == 4;
    if (!((depth == 8) || (depth == 16 && ctype != 3) || (sub_byte && (ctype == 0 || ctype == 3)))) { if (err) *err = "unsupported PNG bit depth"; return false; }
    const int bpp = sub_byte ? 1 : ch * depth / 8;           // filter distance in bytes (1 for packed samples)
    const size_t stride = sub_byte ? ((size_t)w * depth + 7) / 8 : (size_t)w * bpp;
    std::vector<uint8_t> raw;
    raw.reserve((stride + 1) * h);
    if (!inflate(idat.data(), idat.size(), raw) || raw.size() < (stride + 1) * (size_t)h) { if (err) *err = "PNG data does not inflate"; return false; }
    std::vector<uint8_t> img(stride * h);
    for (int y = 0; y < h; y++) {
        const uint8_t ft = raw[(stride + 1) * y];
        const uint8_t *in = &raw[(stride + 1) * y + 1];
        uint8_t *cur = &img[stride * y];
        const uint8_t *up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)bpp) ? up[i - bpp] : 0;
            int v = in[i];
            switch (ft) {
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: { const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
            default: break;
            }
            cur[i] = (uint8_t)v;
        }
    }
    rgb.resize((size_t)w * h * 3);
    if (sub_byte) {
        // packed 1/2/4-bit samples, leftmost pixel in the high-order bits; grey levels scale to 0..255
        // (lodepng's conversion to 8 bit: value * 255 / (2^depth - 1))
        const int maxv = (1 << depth) - 1;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const size_t bit = (size_t)x * depth;
                const int v = (img[stride * y + bit / 8] >> (8 - depth - (int)(bit % 8))) & maxv;
                uint8_t r, g, b;
                if (ctype == 3) { const size_t k = (size_t)v * 3; if (k + 2 < plte.size()) { r = plte[k]; g = plte[k + 1]; b = plte[k + 2]; } else r = g = b = 0; }
                else r = g = b = (uint8_t)(v * 255 / maxv);
                uint8_t *o = &rgb[3 * ((size_t)y * w + x)];
                o[0] = r; o[1] = g; o[2] = b;
            }
        return true;
    }
    const int bs = depth / 8;                  // bytes per sample (16 bit: high byte first)
    for (size_t i = 0; i < (size_t)w * h; i++) {
        const uint8_t *px = &img[i * bpp];
        uint8_t r, g, b;
        if (ctype == 0 || ctype == 4) r = g = b = px[0];
        else if (ctype == 3) { const size_t k = (size_t)px[0] * 3; if (k + 2 < plte.size() + 0 && k + 2 < plte.size()) { r = plte[k]; g = plte[k + 1]; b = plte[k + 2]; } else r = g = b = 0; }
        else { r = px[0]; g = px[bs]; b = px[2 * bs]; }
        rgb[3 * i] = r; rgb[3 * i + 1] = g; rgb[3 * i + 2] = b;
    }
    return true;
}

// LoadPPM, FIN/texture.cpp:33-53 (binary P6, maxval line ignored)
bool load_ppm(const std::vector<uint8_t> &f, int &w, int &h, std::vector<uint8_t> &rgb, std::string *err)
{
    size_t pos = 0;
    auto line = [&](std::string &s) {
        s.clear();
        while (pos < f.size()) { const char c = (char)f[pos++]; if (c == '\n' || c == '\r') break; s += c; }
    };
    std::string s;
    line(s);
    if (s.size() < 2 || s[0] != 'P' || s[1] != '6') { if (err) *err = "not a P6 PPM"; return false; }
    line(s);
    while (!s.empty() && s[0] == '#') line(s);
    if (sscanf(s.c_str(), "%d %d", &w, &h) != 2 || w <= 0 || h <= 0 || (unsigned long long)w * (unsigned long long)h > (1ull << 28)) { if (err) *err = "bad PPM size"; return false; }
    line(s);
    while (!s.empty() && s[0] == '#') line(s);
    rgb.assign((size_t)w * h * 3, 0);
    const size_t n = std::min(rgb.size(), f.size() - std::min(pos, f.size()));
    if (n) memcpy(rgb.data(), &f[pos], n);
    return true;
}

}  // namespace

// RenderImage::ComputeZBufferImage, FIN/include/scene.h:591-613
void ZBufferImage(const float *zbuffer, size_t size, uint8_t *zbufferImg)
{
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
        // int(f * 255) is undefined for NaN (zmax == zmin: 0/0) in the reference; here NaN -> 0
        int c = (f == f) ? int(f * 255) : 0;
        if (c < 0) c = 0;
        if (c > 255) c = 255;
        zbufferImg[i] = (uint8_t)c;
    }
}

// RenderImage::ComputeSampleCountImage, FIN/include/scene.h:615-637
int SampleCountImage(const uint8_t *sampleCount, size_t size, uint8_t *sampleCountImg)
{
    uint8_t smin = 255, smax = 0;
    for (size_t i = 0; i < size; i++) { if (smin > sampleCount[i]) smin = sampleCount[i]; if (smax < sampleCount[i]) smax = sampleCount[i]; }
    if (smax == smin) { for (size_t i = 0; i < size; i++) sampleCountImg[i] = 0; return smax; }
    for (size_t i = 0; i < size; i++) {
        int c = (255 * (sampleCount[i] - smin)) / (smax - smin);
        if (c < 0) c = 0;
        if (c > 255) c = 255;
        sampleCountImg[i] = (uint8_t)c;
    }
    return smax;
}

bool ReadImageRGB(const char *filename, int &w, int &h, std::vector<uint8_t> &rgb, std::string *err)
{
    w = h = 0;
    const size_t len = strlen(filename);
    if (len < 3) { if (err) *err = "bad image file name"; return false; }
    char ext[4] = {(char)tolower(filename[len - 3]), (char)tolower(filename[len - 2]), (char)tolower(filename[len - 1]), 0};
    FILE *fp = fopen(filename, "rb");
    if (!fp) { if (err) *err = std::string("cannot open ") + filename; return false; }
    std::vector<uint8_t> f;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) f.insert(f.end(), buf, buf + n);
    fclose(fp);
    // nothing may be thrown across the C ABI (rt_image_read_rgb, rt_scene_load_xml): a truncated or hostile
    // file can still run an allocation out of memory
    try {
        if (!strcmp(ext, "png")) return load_png(f, w, h, rgb, err);
        if (!strcmp(ext, "ppm")) return load_ppm(f, w, h, rgb, err);
    } catch (const std::exception &e) {
        if (err) *err = std::string("image decode failed: ") + e.what();
        w = h = 0;
        return false;
    }
    if (err) *err = "unsupported image type (png and ppm only, like the reference)";
    return false;
}

}  // namespace rt
