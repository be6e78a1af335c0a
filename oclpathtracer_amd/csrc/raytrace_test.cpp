// raytrace_test.cpp -- the reference's test harness, re-stated against the MI355X shim.
//
// Mirrors, test for test, what test/main.cpp + test/RaytraceTest.cpp of the reference do:
//   DeviceTest fixture           test/TestBase.h:13-58      SetUp/TearDown = init+allocate / deallocate+quit
//   deviceInfo                   test/main.cpp:57-72
//   MemoryAllocation / writeRead / getHostPtr / kernelExecution
//                                test/main.cpp:74-152 (commented out upstream; the natural shim smoke tests)
//   RayCast                      test/RaytraceTest.cpp:202-291  scene load, upload, frame loop, PPM
// through include/pt_adl.hpp, i.e. with the reference's own call sequence.  The render parameters
// the reference hard-codes (512 x 512, 10 000 frames, ../test/cornellbox.bin) are options here.
//
//   raytrace_test [--device N] [--dim 512] [--frames 10000] [--scene cornellbox.bin]
//                 [--out-dir .] [--dump fb.raw] [--no-batch] [--only RayCast]
// Exit code 0 = every check passed.  Own code; no gtest.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/pt_adl.hpp"

using namespace adl;

static int g_failures = 0;
#define IASSERT(x) do { if (!(x)) { std::fprintf(stderr, "IASSERT failed: %s (%s:%d)\n", #x, __FILE__, __LINE__); ++g_failures; } } while (0)

// ---- shared structs: 64-byte records of GenerateColors.cl:12-28 / RaytraceTest.cpp:50-76 ----------
struct float4_t { float x, y, z, w; };
struct int4_t { int x, y, z, w; };
#pragma pack(push, 1)
struct Material { float4_t albedo, emissive; float roughness; int32_t type; char padding[24]; };
struct Triangle { float4_t p1, p2, p3; int32_t id; char padding[12]; };
#pragma pack(pop)
static_assert(sizeof(Material) == 64 && sizeof(Triangle) == 64, "device record layout");
enum { DIFFUSE = 1, SPECULAR = 2 };

static uint32_t f2c(float a)  // RaytraceTest.cpp:78-83
{
    a *= 255;
    int i = (std::isnan(a) || a >= 2147483648.0f || a < -2147483648.0f) ? INT32_MIN : (int)a;  // x86 cvttss2si
    return (uint32_t)(i < 255 ? i : 255);
}

// loadModel (RaytraceTest.cpp:87-198): own reader of the mesh stream; fields the reference leaves
// uninitialised are zero.
static bool loadModel(const char* filepath, std::vector<Triangle>& tBuffer, std::vector<Material>& materialBuffer)
{
    std::ifstream in(filepath, std::ios::binary);
    if (!in) return false;
    std::vector<char> blob((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    size_t off = 0;
    auto take32 = [&](void* dst) -> bool {
        if (off + 4 > blob.size()) return false;
        std::memcpy(dst, &blob[off], 4);
        off += 4;
        return true;
    };
    int32_t nMesh = 0;
    if (!take32(&nMesh) || nMesh < 0) return false;
    int32_t id = 0;
    for (int32_t i = 0; i < nMesh; ++i) {
        int32_t nf = 0, nv = 0;
        float tag = 0.f;
        if (!take32(&nf) || !take32(&tag) || nf < 0 || off + (size_t)nf * 16 > blob.size()) return false;
        std::vector<int4_t> idx((size_t)nf);
        if (nf) std::memcpy(idx.data(), &blob[off], (size_t)nf * 16);
        off += (size_t)nf * 16;
        if (!take32(&nv) || nv < 0 || off + (size_t)nv * 16 > blob.size()) return false;
        std::vector<float4_t> vtx((size_t)nv);
        if (nv) std::memcpy(vtx.data(), &blob[off], (size_t)nv * 16);
        off += (size_t)nv * 16;

        Material mat;
        std::memset(&mat, 0, sizeof mat);
        mat.type = DIFFUSE;
        if (tag != 0.5f) {  // the light (:147-151)
            mat.emissive = { 30.0f, 30.0f, 30.0f, 1.0f };
            mat.albedo = { 1.0f, 1.0f, 1.0f, 1.0f };
        } else {
            mat.emissive = { 0.0f, 0.0f, 0.0f, 1.0f };
        }
        if (i == 0 || i == 1 || i == 2) mat.albedo = { 0.7f, 0.7f, 0.7f, 1.0f };
        if (i == 3) mat.albedo = { 0.6f, 0.0f, 0.0f, 1.0f };
        if (i == 4) mat.albedo = { 0.0f, 0.6f, 0.0f, 1.0f };
        if (i == 5) {
            mat.albedo = { 0.5f, 0.35f, 0.05f, 0.0f };
            mat.roughness = 0.008f;
            mat.type = SPECULAR;
        }
        for (int32_t j = 0; j < nf; ++j) {
            const int4_t q = idx[(size_t)j];
            const int v[4] = { q.x, q.y, q.z, q.w };
            float4_t p[4];
            for (int k = 0; k < 4; ++k) {
                if (v[k] < 0 || v[k] >= nv) return false;
                p[k] = { vtx[(size_t)v[k]].x, vtx[(size_t)v[k]].y, vtx[(size_t)v[k]].z, 0.0f };
            }
            Triangle t1, t2;
            std::memset(&t1, 0, sizeof t1);
            std::memset(&t2, 0, sizeof t2);
            t1.p1 = p[0]; t1.p2 = p[1]; t1.p3 = p[2]; t1.id = id;  // (a,b,c)
            t2.p1 = p[2]; t2.p2 = p[3]; t2.p3 = p[0]; t2.id = id;  // (c,d,a)
            tBuffer.push_back(t1);
            tBuffer.push_back(t2);
            materialBuffer.push_back(mat);
            ++id;
        }
    }
    return tBuffer.size() / 2 == materialBuffer.size();
}

// ---- fixture -------------------------------------------------------------------------------------
struct Options {
    int deviceIdx = 0, dim = 512, frames = 10000;
    bool batch = true;
    std::string scene = "cornellbox.bin", outDir = ".", dump, only;
};

struct DeviceTest {
    Device* m_d = nullptr;
    bool SetUp(const Options& o)
    {
        if (!adl::init(TYPE_HIP)) { std::fprintf(stderr, "adl::init failed: %s\n", pt_last_error()); return false; }
        DeviceUtils::Config cfg;
        cfg.m_deviceIdx = o.deviceIdx;
        m_d = DeviceUtils::allocate(TYPE_HIP, cfg);
        IASSERT(m_d != 0);
        if (m_d && !o.batch) pt_device_set_option(m_d->m_handle, PT_OPT_BATCH_FRAMES, 0);
        return m_d != 0;
    }
    void TearDown()
    {
        DeviceUtils::deallocate(m_d);
        adl::quit(TYPE_HIP);
    }
    void getFilePath(const char* dir, const char* prefix, const char* ext, char* dst, size_t n)
    {
        char t[128];
        m_d->getDeviceVersion(t);
        for (char* c = t; *c; ++c)
            if (*c == ' ' || *c == '/' || *c == ':') *c = '_';
        std::snprintf(dst, n, "%s/%s_%s.%s", dir, prefix, t, ext);
    }
};

static void test_deviceInfo(DeviceTest& f)
{
    char t[128];
    f.m_d->getDeviceName(t);    std::printf("Device Name:    %s\n", t); IASSERT(t[0] != 0);
    f.m_d->getBoardName(t);     std::printf("Board Name:     %s\n", t);
    f.m_d->getDeviceVendor(t);  std::printf("Device Vendor:  %s\n", t); IASSERT(t[0] != 0);
    f.m_d->getDeviceVersion(t); std::printf("Device Version: %s\n", t); IASSERT(std::strstr(t, "gfx950") != 0);
    std::printf("Max Allocation Size: %3.2fMB\n", f.m_d->getMaxAllocationSize() / 1024.f / 1024.f);
    IASSERT(f.m_d->getMaxAllocationSize() > 0);
}

static void test_MemoryAllocation(DeviceTest& f)
{
    const adlu64 n = 256ull << 20;  // 256 MiB (the upstream test grabs 90 % of max alloc; bounded here)
    {
        Buffer<char> b(f.m_d, n);
        IASSERT(b.m_ptr != 0 && b.getSize() == n);
        IASSERT(f.m_d->getUsedMemory() >= n);
    }
    IASSERT(f.m_d->getUsedMemory() == 0);
    IASSERT(f.m_d->getPeakMemory() >= n);
}

static void test_writeRead(DeviceTest& f)
{
    const int n = 128;
    int host[n], back[n];
    for (int i = 0; i < n; ++i) { host[i] = i * 3 + 1; back[i] = -1; }
    Buffer<int> b(f.m_d, n);
    b.write(host, n);
    DeviceUtils::waitForCompletion(f.m_d);
    b.read(back, n);
    DeviceUtils::waitForCompletion(f.m_d);
    for (int i = 0; i < n; ++i) IASSERT(back[i] == host[i]);
    Buffer<int> c(f.m_d, n);
    c.write(b, n);  // device-to-device
    int back2[n];
    c.read(back2, 64, 64);  // offset read
    DeviceUtils::waitForCompletion(f.m_d);
    for (int i = 0; i < 64; ++i) IASSERT(back2[i] == host[64 + i]);
}

static void test_getHostPtr(DeviceTest& f)
{
    const int n = 1024;
    Buffer<float> b(f.m_d, n);
    float* p = b.getHostPtr();
    DeviceUtils::waitForCompletion(f.m_d);
    IASSERT(p != 0);
    for (int i = 0; i < n; ++i) p[i] = (float)i * 0.5f;
    b.returnHostPtr(p);
    DeviceUtils::waitForCompletion(f.m_d);
    p = b.getHostPtr(-1, true);
    for (int i = 0; i < n; ++i) IASSERT(p[i] == (float)i * 0.5f);
    b.returnHostPtr(p);
    DeviceUtils::waitForCompletion(f.m_d);
}

static void test_kernelExecution(DeviceTest& f)
{
    const int n = 1000;
    Buffer<int> b(f.m_d, n);
    Kernel* k = f.m_d->getKernel("../test/PtShimTest", "FillKernel");
    IASSERT(k != 0);
    IASSERT(f.m_d->getKernel("../test/ClKernels/NoSuchKernel", "Nope") == 0);  // missing kernel -> 0
    if (!k) return;
    BufferInfo bInfo[] = { BufferInfo(&b) };
    Launcher launcher(f.m_d, k);
    launcher.setBuffers(bInfo, 1);
    int value = 42;
    launcher.setConst(value);
    SyncObject sync(f.m_d);
    launcher.launch1D(n, 64, &sync);
    DeviceUtils::waitForCompletion(&sync);
    IASSERT(DeviceUtils::isComplete(&sync));
    std::vector<int> host((size_t)n, 0);
    b.read(host.data(), n);
    DeviceUtils::waitForCompletion(f.m_d);
    for (int i = 0; i < n; ++i) IASSERT(host[(size_t)i] == 42);
}

// TEST_F(DeviceTest, RayCast): RaytraceTest.cpp:202-291 with dimension / frame count as options
static void test_RayCast(DeviceTest& f, const Options& o)
{
    Device* m_d = f.m_d;
    std::vector<Triangle> triangles;
    std::vector<Material> materials;
    triangles.reserve(18 * 2);
    materials.reserve(18);
    if (!loadModel(o.scene.c_str(), triangles, materials)) {
        std::printf("Error loading model !!\n");
        ++g_failures;
        return;
    }

    const int dimension = o.dim;
    Buffer<float4_t> frameBuff(m_d, (adlu64)dimension * dimension);
    // (the reference passes a byte count as nElems here, over-allocating 64x: not reproduced)
    Buffer<Triangle>* tBuffer = new Buffer<Triangle>(m_d, triangles.size());
    Buffer<Material>* materialBuffer = new Buffer<Material>(m_d, materials.size());

    Triangle* tb = tBuffer->getHostPtr();
    Material* mb = materialBuffer->getHostPtr();
    DeviceUtils::waitForCompletion(m_d);
    for (size_t i = 0; i < triangles.size(); i++) tb[i] = triangles[i];
    for (size_t i = 0; i < materials.size(); i++) mb[i] = materials[i];
    tBuffer->returnHostPtr(tb);
    materialBuffer->returnHostPtr(mb);
    DeviceUtils::waitForCompletion(m_d);

    auto t0 = std::chrono::steady_clock::now();
    unsigned int frameCount = 0;
    while (frameCount != (unsigned)o.frames) {
        int4_t res;
        res.x = dimension; res.y = dimension; res.z = (int)frameCount++; res.w = 0;
        BufferInfo bInfo[] = { BufferInfo(tBuffer), BufferInfo(materialBuffer), BufferInfo(&frameBuff) };
        Launcher launcher(m_d, m_d->getKernel(SELECT_KERNELPATH1(m_d, "../test/", "GenerateColors"), "GenerateColors"));
        launcher.setBuffers(bInfo, sizeof(bInfo) / sizeof(BufferInfo));
        launcher.setConst(res);
        launcher.launch1D(dimension * dimension);
        DeviceUtils::waitForCompletion(m_d);
    }

    // Save rendering to file
    {
        float4_t* h = frameBuff.getHostPtr();
        DeviceUtils::waitForCompletion(m_d);
        double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("RayCast: %d x %d x %d frames in %.3f s = %.1f Msamples/s (frame loop + readback)\n", dimension, dimension,
                    o.frames, secs, (double)dimension * dimension * o.frames / secs / 1e6);
        IASSERT(h != 0);
        char path[512];
        f.getFilePath(o.outDir.c_str(), "rayCastAo", "ppm", path, sizeof path);
        FILE* fp = std::fopen(path, "w");
        IASSERT(fp != 0);
        if (fp && h) {
            std::fprintf(fp, "P3\n%d %d\n%d\n", dimension, dimension, 255);
            for (int i = 0; i < dimension * dimension; i++) {
                float4_t v = h[i];
                std::fprintf(fp, "%d %d %d ", f2c(std::sqrt(v.x)), f2c(std::sqrt(v.y)), f2c(std::sqrt(v.z)));
            }
            std::fclose(fp);
            std::printf("wrote %s\n", path);
        }
        if (h && !o.dump.empty()) {
            FILE* fd = std::fopen(o.dump.c_str(), "wb");
            IASSERT(fd != 0);
            if (fd) {
                std::fwrite(h, sizeof(float4_t), (size_t)dimension * dimension, fd);
                std::fclose(fd);
            }
        }
        if (h && o.frames > 0)
            for (int i = 0; i < dimension * dimension; i += 97) IASSERT(h[i].w == 1.0f);  // gammaCorrect sets w = 1 (:293)
        DeviceUtils::waitForCompletion(m_d);
    }
    delete tBuffer;          // the reference leaks these two (:222-223)
    delete materialBuffer;
}

int main(int argc, char** argv)
{
    Options o;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--device") o.deviceIdx = std::atoi(next());
        else if (a == "--dim") o.dim = std::atoi(next());
        else if (a == "--frames") o.frames = std::atoi(next());
        else if (a == "--scene") o.scene = next();
        else if (a == "--out-dir") o.outDir = next();
        else if (a == "--dump") o.dump = next();
        else if (a == "--only") o.only = next();
        else if (a == "--no-batch") o.batch = false;
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if (o.dim < 1 || o.frames < 0) { std::fprintf(stderr, "bad --dim/--frames\n"); return 2; }
    struct { const char* name; int kind; } tests[] = { { "initialize", 0 }, { "deviceInfo", 1 }, { "MemoryAllocation", 2 }, { "writeRead", 3 },
                                                       { "getHostPtr", 4 }, { "kernelExecution", 5 }, { "RayCast", 6 } };
    for (auto& t : tests) {
        if (!o.only.empty() && o.only != t.name) continue;
        std::printf("[ RUN      ] DeviceTest.%s\n", t.name);
        int before = g_failures;
        DeviceTest f;
        if (!f.SetUp(o)) { std::printf("[  FAILED  ] DeviceTest.%s (no device)\n", t.name); return 1; }
        switch (t.kind) {
        case 1: test_deviceInfo(f); break;
        case 2: test_MemoryAllocation(f); break;
        case 3: test_writeRead(f); break;
        case 4: test_getHostPtr(f); break;
        case 5: test_kernelExecution(f); break;
        case 6: test_RayCast(f, o); break;
        default: break;
        }
        f.TearDown();
        std::printf(g_failures == before ? "[       OK ] DeviceTest.%s\n" : "[  FAILED  ] DeviceTest.%s\n", t.name);
    }
    return g_failures ? 1 : 0;
}
