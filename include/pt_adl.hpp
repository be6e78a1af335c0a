// pt_adl.hpp -- header-only C++ facade with the reference's "Adl" names over the C ABI of
// libptshim.so (include/pt_shim.h).  It lets a RaytraceTest-shaped harness
// (reference: test/RaytraceTest.cpp:202-291, test/TestBase.h:13-58) compile against the MI355X
// shim with the calls it already makes:
//
//   adl::init / adl::quit                                   Adl/Adl.h:96-98
//   adl::DeviceUtils::{Config, allocate, deallocate, waitForCompletion, getNDevices, flush}
//                                                            Adl/Adl.h:100-131
//   adl::Device::{getDeviceName, getBoardName, getDeviceVendor, getDeviceVersion,
//                 getMaxAllocationSize, getKernel, toggleProfiling, ...}   Adl/Adl.h:139-194
//   adl::Buffer<T>(device, nElems), getHostPtr, returnHostPtr, write, read  Adl/Adl.h:203-265
//   adl::BufferInfo, adl::Launcher{setBuffers, setConst, launch1D, launch2D} Adl/AdlKernel.h:59-202
//   adl::SyncObject                                         Adl/AdlKernel.h:45-54
//
// Same argument meaning and the same "null / zero on failure" behaviour.
// Differences, all deliberate:
//   * the only backend is TYPE_HIP (TYPE_CL is an alias so reference call sites compile);
//     TYPE_HOST is not offered -- the reference's host backend cannot launch kernels
//     (Adl/AdlKernel.inl:101-106) and this library has no CPU path;
//   * allocate() returns NULL when no MI355X is usable (the reference returns a non-null device
//     with isValid()==false, Adl/CL/AdlCL.cpp:148-151): the hot path must fail loudly;
//   * kernels come from a static registry compiled for gfx950; nothing is built at run time.
#ifndef PT_ADL_HPP
#define PT_ADL_HPP

#include <cstdio>
#include <cstring>

#include "pt_shim.h"

namespace adl {

typedef unsigned long long adlu64;

enum DeviceType { TYPE_CL = 0, TYPE_HIP = 0, TYPE_DX11 = 1, TYPE_METAL = 2, TYPE_VULKAN = 3, TYPE_HOST = 4 };

#define ADL_SUCCESS 0
#define ADL_FAILURE 1
#define ADL_DEFAULT_LOCAL_SIZE_1D 64
#define ADL_DEFAULT_LOCAL_SIZE_2D 8
// the reference's SELECT_KERNELPATH1(device, dir, name) yields "<dir>ClKernels/<name>" for CL
#define SELECT_KERNELPATH1(dev, dir, name) dir "ClKernels/" name

inline void adlLog(const char* what) { std::fprintf(stderr, "[adl/hip] %s: %s\n", what, pt_last_error()); }

inline bool init(DeviceType type) { return type == TYPE_HIP && pt_init() == PT_OK; }
inline void quit(DeviceType type) { if (type == TYPE_HIP) pt_quit(); }

struct Kernel {
    DeviceType m_type;
    void* m_kernel;  // pt_kernel_t, owned by the device
    const char* m_funcName;
};

struct SyncObject;

struct Device {
    enum ProfileType { PROFILE_NON = 0, PROFILE_RETURN_TIME = 1 << 1, PROFILE_WRITE_FILE = 1 << 2 };
    enum { MAX_KERNELS = 8 };

    explicit Device(DeviceType type) : m_type(type), m_handle(0), m_nKernels(0), m_enableProfiling(0) {}

    bool isValid() const { return m_handle != 0; }
    void getDeviceName(char out[128]) const { info(PT_INFO_NAME, out); }
    void getBoardName(char out[128]) const { info(PT_INFO_BOARD, out); }
    void getDeviceVendor(char out[128]) const { info(PT_INFO_VENDOR, out); }
    void getDeviceVersion(char out[128]) const { info(PT_INFO_VERSION, out); }
    adlu64 getUsedMemory() const { return pt_device_used_memory(m_handle); }
    adlu64 getPeakMemory() const { return pt_device_peak_memory(m_handle); }
    adlu64 getTotalMemory() const { return pt_device_mem_size(m_handle); }
    adlu64 getMemSize() const { return pt_device_mem_size(m_handle); }
    adlu64 getMaxAllocationSize() const { return pt_device_max_alloc(m_handle); }
    DeviceType getType() const { return m_type; }
    void waitForCompletion() const { if (pt_sync(m_handle) != PT_OK) adlLog("waitForCompletion"); }
    void flush() const { if (pt_flush(m_handle) != PT_OK) adlLog("flush"); }

    void toggleProfiling(ProfileType type)
    {
        m_enableProfiling |= type;
        if (type == PROFILE_NON) m_enableProfiling = PROFILE_NON;
        pt_device_set_option(m_handle, PT_OPT_PROFILE_RETURN_TIME, (m_enableProfiling & PROFILE_RETURN_TIME) ? 1 : 0);
    }

    // Device::getKernel: 0 when the kernel does not exist (Adl/AdlKernel.cpp:176-181)
    Kernel* getKernel(const char* fileName, const char* funcName, const char* /*option*/ = 0) const
    {
        pt_kernel_t k = 0;
        if (pt_kernel_get(m_handle, fileName, funcName, &k) != PT_OK) return 0;
        for (int i = 0; i < m_nKernels; ++i)
            if (m_kernels[i].m_kernel == (void*)k) return &m_kernels[i];
        if (m_nKernels == MAX_KERNELS) return 0;
        Kernel& out = m_kernels[m_nKernels++];
        out.m_type = m_type;
        out.m_kernel = (void*)k;
        out.m_funcName = funcName;
        return &out;
    }

    DeviceType m_type;
    pt_device_t m_handle;
    mutable Kernel m_kernels[MAX_KERNELS];
    mutable int m_nKernels;
    unsigned int m_enableProfiling;

private:
    void info(int kind, char out[128]) const
    {
        out[0] = 0;
        if (pt_device_info(m_handle, kind, out) != PT_OK) adlLog("device info");
    }
};

struct SyncObject {
    explicit SyncObject(const Device* device) : m_device(device), m_ptr(0)
    {
        pt_event_t e = 0;
        if (pt_event_create(device->m_handle, &e) == PT_OK) m_ptr = e;
        else adlLog("SyncObject");
    }
    ~SyncObject() { if (m_ptr) pt_event_destroy((pt_event_t)m_ptr); }
    const Device* m_device;
    void* m_ptr;
private:
    SyncObject(const SyncObject&);
    SyncObject& operator=(const SyncObject&);
};

inline pt_event_t adlEvent(SyncObject* s) { return s ? (pt_event_t)s->m_ptr : 0; }

class DeviceUtils {
public:
    struct Config {
        enum DeviceType { DEVICE_GPU, DEVICE_CPU };
        Config() : m_type(DEVICE_GPU), m_deviceIdx(0) {}
        DeviceType m_type;
        int m_deviceIdx;
    };

    static int getNDevices(adl::DeviceType type) { return type == TYPE_HIP ? pt_device_count() : 0; }
    static int getNCUs(const Device* d) { return pt_device_num_cus(d->m_handle); }

    static Device* allocate(adl::DeviceType type, Config cfg = Config())
    {
        if (type != TYPE_HIP) return 0;  // unknown backend -> 0 (Adl/Adl.cpp:188-189)
        pt_device_t h = 0;
        if (pt_device_create(cfg.m_deviceIdx, &h) != PT_OK) { adlLog("DeviceUtils::allocate"); return 0; }
        Device* d = new Device(type);
        d->m_handle = h;
        return d;
    }
    static void deallocate(Device* d)
    {
        if (!d) return;
        if (pt_device_destroy(d->m_handle) != PT_OK) adlLog("DeviceUtils::deallocate");  // ref: ADLASSERT(used == 0)
        delete d;
    }
    static void waitForCompletion(const Device* d) { d->waitForCompletion(); }
    static void waitForCompletion(const SyncObject* s) { if (s && s->m_ptr) pt_event_wait((pt_event_t)s->m_ptr); }
    static bool isComplete(const SyncObject* s) { return !s || !s->m_ptr || pt_event_is_complete((pt_event_t)s->m_ptr) == 1; }
    static void flush(const Device* d) { d->flush(); }
    static adlu64 getExecutionTimeNanoseconds(const SyncObject* s)
    {
        uint64_t ns = 0;
        if (s && s->m_ptr) pt_event_elapsed_ns((pt_event_t)s->m_ptr, &ns);
        return ns;
    }
};

struct BufferBase { enum BufferType { BUFFER }; };

template <typename T>
struct Buffer : public BufferBase {
    Buffer() : m_device(0), m_size(0), m_ptr(0), m_handle(0), m_mapped(0) {}
    Buffer(const Device* device, adlu64 nElems, BufferType = BUFFER) : m_device(0), m_size(0), m_ptr(0), m_handle(0), m_mapped(0)
    {
        allocate(device, nElems);
    }
    virtual ~Buffer() { release(); }

    // Buffer<T>::allocate: failure leaves m_size = 0, m_ptr = 0 and logs (Adl/CL/AdlCL.inl:190-197)
    void allocate(const Device* device, adlu64 nElems, BufferType = BUFFER)
    {
        release();
        m_device = device;
        pt_buffer_t b = 0;
        if (pt_buffer_alloc(device->m_handle, (size_t)(nElems * sizeof(T)), &b) != PT_OK) {
            adlLog("HIP Memory Allocation Failure");
            return;
        }
        m_handle = b;
        m_size = nElems;
        m_ptr = (T*)pt_buffer_address(b);  // the address as a value (the reference's m_ptr is the opaque cl_mem); getInternalObject() licenses access
    }
    // Buffer<T>::setRawPtr: adopt caller-owned device memory (Adl/Adl.h:214)
    void setRawPtr(const Device* device, T* ptr, adlu64 size, BufferType = BUFFER)
    {
        release();
        m_device = device;
        pt_buffer_t b = 0;
        if (pt_buffer_wrap(device->m_handle, ptr, (size_t)(size * sizeof(T)), &b) != PT_OK) { adlLog("setRawPtr"); return; }
        m_handle = b;
        m_size = size;
        m_ptr = ptr;
    }
    void write(const T* hostSrc, adlu64 nElems, adlu64 dstOffsetNElems = 0, SyncObject* sync = 0)
    {
        if (pt_buffer_write(m_handle, hostSrc, (size_t)(nElems * sizeof(T)), (size_t)(dstOffsetNElems * sizeof(T)), adlEvent(sync)) != PT_OK)
            adlLog("Buffer::write");
    }
    void read(T* hostDst, adlu64 nElems, adlu64 srcOffsetNElems = 0, SyncObject* sync = 0) const
    {
        if (pt_buffer_read(m_handle, hostDst, (size_t)(nElems * sizeof(T)), (size_t)(srcOffsetNElems * sizeof(T)), adlEvent(sync)) != PT_OK)
            adlLog("Buffer::read");
    }
    void write(const Buffer<T>& src, adlu64 nElems, adlu64 dstOffsetNElems = 0, SyncObject* sync = 0)
    {
        if (pt_buffer_copy(m_handle, src.m_handle, (size_t)(nElems * sizeof(T)), (size_t)(dstOffsetNElems * sizeof(T)), 0, adlEvent(sync)) != PT_OK)
            adlLog("Buffer::write(Buffer)");
    }
    void read(Buffer<T>& dst, adlu64 nElems, adlu64 offsetNElems = 0, SyncObject* sync = 0) const
    {
        if (pt_buffer_copy(dst.m_handle, m_handle, (size_t)(nElems * sizeof(T)), 0, (size_t)(offsetNElems * sizeof(T)), adlEvent(sync)) != PT_OK)
            adlLog("Buffer::read(Buffer)");
    }
    // non-blocking by default, like clEnqueueMapBuffer(blocking = CL_FALSE): wait before use
    T* getHostPtr(adlu64 size = (adlu64)-1, bool blocking = false) const
    {
        size_t bytes = size == (adlu64)-1 ? (size_t)-1 : (size_t)(size * sizeof(T));
        void* p = pt_buffer_map(m_handle, bytes, blocking ? 1 : 0);
        if (!p) adlLog("Buffer::getHostPtr");
        m_mapped = p;
        return (T*)p;
    }
    void returnHostPtr(T* ptr) const
    {
        if (pt_buffer_unmap(m_handle, ptr) != PT_OK) adlLog("Buffer::returnHostPtr");
        m_mapped = 0;
    }
    adlu64 getSize() const { return m_size; }
    // the device pointer for code that touches the memory itself: submits deferred frames and ends frame batching for this buffer
    void* getInternalObject() { return m_handle ? pt_buffer_device_ptr(m_handle) : (void*)m_ptr; }
    DeviceType getType() const { return m_device->m_type; }

    void release()
    {
        if (m_handle) pt_buffer_free(m_handle);
        m_handle = 0;
        m_size = 0;
        m_ptr = 0;
        m_mapped = 0;
    }

    const Device* m_device;
    adlu64 m_size;
    T* m_ptr;
    pt_buffer_t m_handle;
    mutable void* m_mapped;

private:
    Buffer(const Buffer&);
    Buffer& operator=(const Buffer&);
};

struct BufferInfo {
    BufferInfo() : m_buffer(0), m_handle(0), m_isReadOnly(false) {}
    template <typename T>
    BufferInfo(Buffer<T>* buff, bool isReadOnly = false) : m_buffer(buff), m_handle(buff->m_handle), m_isReadOnly(isReadOnly) {}
    template <typename T>
    BufferInfo(const Buffer<T>* buff, bool isReadOnly = false) : m_buffer((void*)buff), m_handle(buff->m_handle), m_isReadOnly(isReadOnly) {}
    void* m_buffer;
    pt_buffer_t m_handle;
    bool m_isReadOnly;
};

class Launcher {
public:
    enum { MAX_ARG_SIZE = PT_MAX_ARG_SIZE, MAX_ARG_COUNT = PT_MAX_ARG_COUNT };

    Launcher(const Device* dd, const Kernel* kernel) : m_deviceData(dd), m_kernel(kernel), m_idx(0) {}
    Launcher(const Device* dd, const char* fileName, const char* funcName, const char* option = 0)
        : m_deviceData(dd), m_kernel(dd->getKernel(fileName, funcName, option)), m_idx(0) {}

    void setBuffers(BufferInfo* buffInfo, int n)
    {
        for (int i = 0; i < n && m_idx < MAX_ARG_COUNT; ++i) {
            pt_launch_arg& a = m_args[m_idx++];
            a.is_buffer = 1;
            a.read_only = buffInfo[i].m_isReadOnly ? 1 : 0;
            a.size = 0;
            a.buffer = buffInfo[i].m_handle;
        }
    }
    template <typename T>
    void setConst(const T& consts) { setConst(&consts, sizeof(T)); }
    void setConst(const void* consts, size_t byteCount)
    {
        if (byteCount > (size_t)MAX_ARG_SIZE || m_idx >= MAX_ARG_COUNT) { std::fprintf(stderr, "[adl/hip] setConst: argument too large\n"); return; }
        pt_launch_arg& a = m_args[m_idx++];
        a.is_buffer = 0;
        a.read_only = 0;
        a.size = byteCount;
        a.buffer = 0;
        std::memcpy(a.data, consts, byteCount);
    }
    float launch1D(int numThreads, int localSize = ADL_DEFAULT_LOCAL_SIZE_1D, SyncObject* sync = 0)
    {
        return launch2D(numThreads, 1, localSize, 1, sync);
    }
    float launch2D(int numThreadsX, int numThreadsY, int localSizeX = ADL_DEFAULT_LOCAL_SIZE_2D,
                   int localSizeY = ADL_DEFAULT_LOCAL_SIZE_2D, SyncObject* sync = 0)
    {
        float ms = 0.f;
        pt_kernel_t k = m_kernel ? (pt_kernel_t)m_kernel->m_kernel : 0;  // a null kernel is reported, not dereferenced
        if (pt_launch_2d(m_deviceData->m_handle, k, m_args, m_idx, numThreadsX, numThreadsY, localSizeX, localSizeY, adlEvent(sync), &ms) != PT_OK)
            adlLog("Launcher::launch2D");
        return ms;
    }

    const Device* m_deviceData;
    const Kernel* m_kernel;
    int m_idx;
    pt_launch_arg m_args[MAX_ARG_COUNT];
};

}  // namespace adl

#endif  // PT_ADL_HPP
