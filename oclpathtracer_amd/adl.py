"""Python mirror of the subset of the reference's "Adl" host API that RaytraceTest drives.

Same names, argument meaning and failure behaviour as the reference, layered
on the C ABI of ``libptshim.so`` so that a parity test reads like ``test/RaytraceTest.cpp``:

    adl.init(adl.TYPE_HIP)
    dev = adl.DeviceUtils.allocate(adl.TYPE_HIP, cfg)
    fb = adl.Buffer(dev, W * H, adl.float4)
    k = dev.getKernel("../test/ClKernels/GenerateColors", "GenerateColors")
    l = adl.Launcher(dev, k); l.setBuffers([...]); l.setConst(res); l.launch1D(W * H)
    adl.DeviceUtils.waitForCompletion(dev)

Reference: Adl/Adl.h:96-131 (init/quit/DeviceUtils), :139-194 (Device), :203-265 (Buffer),
Adl/AdlKernel.h:45-54 (SyncObject), :59-69 (BufferInfo), :121-202 (Launcher).
The only backend is the MI355X HIP shim (``TYPE_HIP``); ``TYPE_CL`` is accepted as an alias
so reference-shaped call sites run unchanged.  ``TYPE_HOST`` is not offered: the reference's
host backend cannot launch kernels (Adl/AdlKernel.inl:101-106) and this package has no CPU path.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import numpy as np

from . import shim

TYPE_CL = 0   # alias of TYPE_HIP: "the GPU backend"
TYPE_HIP = 0
TYPE_HOST = 4

ADL_DEFAULT_LOCAL_SIZE_1D = 64  # Adl/AdlKernel.h:71
ADL_DEFAULT_LOCAL_SIZE_2D = 8

float4 = np.dtype((np.float32, 4))
int4 = np.dtype((np.int32, 4))


def init(device_type: int = TYPE_HIP) -> bool:
    """adl::init (Adl/Adl.cpp:39-58): True when the backend is usable."""
    if device_type != TYPE_HIP:
        return False
    try:
        return shim.load().pt_init() == shim.PT_OK
    except (shim.ShimError, OSError):
        return False


def quit(device_type: int = TYPE_HIP) -> None:  # noqa: A001 - reference name
    if device_type == TYPE_HIP:
        shim.load().pt_quit()


class Config:
    """DeviceUtils::Config (Adl/Adl.h:103-119); only m_deviceIdx matters here."""

    def __init__(self, m_deviceIdx: int = 0):
        self.m_deviceIdx = m_deviceIdx


class Kernel:
    """adl::Kernel (Adl/AdlKernel.h:17-22): an opaque handle owned by the device."""

    def __init__(self, handle, func_name: str):
        self.m_kernel = handle
        self.m_funcName = func_name


class Device:
    """adl::Device over a pt_device_t."""

    PROFILE_NON = 0
    PROFILE_RETURN_TIME = 1 << 1

    def __init__(self, handle):
        self._h = handle
        self.m_type = TYPE_HIP
        self._lib = shim.load()

    # -- validity / info (Adl/Adl.h:153-177)
    def isValid(self) -> bool:
        return bool(self._h)

    def _info(self, kind: int) -> str:
        buf = ctypes.create_string_buffer(128)
        shim.check(self._lib.pt_device_info(self._h, kind, buf))
        return buf.value.decode()

    def getDeviceName(self) -> str:
        return self._info(shim.PT_INFO_NAME)

    def getBoardName(self) -> str:
        return self._info(shim.PT_INFO_BOARD)

    def getDeviceVendor(self) -> str:
        return self._info(shim.PT_INFO_VENDOR)

    def getDeviceVersion(self) -> str:
        return self._info(shim.PT_INFO_VERSION)

    def getMaxAllocationSize(self) -> int:
        return self._lib.pt_device_max_alloc(self._h)

    def getMemSize(self) -> int:
        return self._lib.pt_device_mem_size(self._h)

    def getUsedMemory(self) -> int:
        return self._lib.pt_device_used_memory(self._h)

    def getPeakMemory(self) -> int:
        return self._lib.pt_device_peak_memory(self._h)

    def getWorkspaceMemory(self) -> int:
        """Device memory the handle holds for itself (staging ring, prepared scene, LBVH, masks): no Adl counterpart."""
        return self._lib.pt_device_workspace_memory(self._h)

    def reserveStaging(self, nbytes: int = 0) -> None:
        """Size the renderer's radiance staging ring once (0 = default); renders then never allocate (pt_device_reserve_staging)."""
        shim.check(self._lib.pt_device_reserve_staging(self._h, int(nbytes)))

    def waitStream(self, hip_stream: int) -> None:
        """Later work of this device starts after what is enqueued on ``hip_stream`` (device-side wait; ``torch.cuda.Stream.cuda_stream``;
        torch's default stream reports 0 = the legacy default stream, PT_STREAM_LEGACY)."""
        shim.check(self._lib.pt_device_wait_stream(self._h, ctypes.c_void_p(int(hip_stream) or shim.PT_STREAM_LEGACY)))

    def waitHipEvent(self, hip_event: int) -> None:
        """Later work of this device starts after a recorded hipEvent_t (``torch.cuda.Event.cuda_event``)."""
        shim.check(self._lib.pt_device_wait_hip_event(self._h, ctypes.c_void_p(int(hip_event))))

    def toggleProfiling(self, profile_type: int) -> None:
        on = 1 if (profile_type & Device.PROFILE_RETURN_TIME) else 0
        shim.check(self._lib.pt_device_set_option(self._h, shim.PT_OPT_PROFILE_RETURN_TIME, on))

    def setOption(self, option: int, value: int) -> None:
        shim.check(self._lib.pt_device_set_option(self._h, option, value))

    def getKernel(self, fileName: str, funcName: str, option: Optional[str] = None) -> Optional[Kernel]:
        """Device::getKernel (Adl/CL/AdlCL.cpp:490-493): None when no such kernel exists
        (the reference returns 0 for a missing file, Adl/AdlKernel.cpp:176-181)."""
        out = ctypes.c_void_p()
        rc = self._lib.pt_kernel_get(self._h, fileName.encode(), funcName.encode(), ctypes.byref(out))
        if rc == shim.PT_ERR_NOT_FOUND:
            return None
        shim.check(rc)
        return Kernel(out.value, funcName)

    def waitForCompletion(self) -> None:
        shim.check(self._lib.pt_sync(self._h))

    def flush(self) -> None:
        shim.check(self._lib.pt_flush(self._h))


class DeviceUtils:
    """DeviceUtils (Adl/Adl.h:100-131, Adl/Adl.cpp:84-232)."""

    Config = Config

    @staticmethod
    def getNDevices(device_type: int = TYPE_HIP) -> int:
        return shim.load().pt_device_count() if device_type == TYPE_HIP else 0

    @staticmethod
    def getNCUs(device: Device) -> int:
        return shim.load().pt_device_num_cus(device._h)

    @staticmethod
    def allocate(device_type: int = TYPE_HIP, cfg: Optional[Config] = None) -> Optional[Device]:
        """Returns None for an unknown backend (Adl/Adl.cpp:188-189).  Unlike the reference,
        which hands back an invalid non-null device when no GPU exists (AdlCL.cpp:148-151),
        a missing MI355X raises ShimError: the hot path must fail loudly."""
        if device_type != TYPE_HIP:
            return None
        cfg = cfg or Config()
        out = ctypes.c_void_p()
        shim.check(shim.load().pt_device_create(cfg.m_deviceIdx, ctypes.byref(out)))
        return Device(out.value)

    @staticmethod
    def deallocate(device: Device) -> None:
        shim.check(shim.load().pt_device_destroy(device._h))
        device._h = None

    @staticmethod
    def waitForCompletion(obj) -> None:
        obj.waitForCompletion()

    @staticmethod
    def isComplete(sync: "SyncObject") -> bool:
        return sync.isComplete()

    @staticmethod
    def flush(device: Device) -> None:
        device.flush()


class SyncObject:
    """adl::SyncObject (Adl/AdlKernel.h:45-54)."""

    def __init__(self, device: Device):
        self.m_device = device
        out = ctypes.c_void_p()
        shim.check(device._lib.pt_event_create(device._h, ctypes.byref(out)))
        self._h = out.value

    def waitForCompletion(self) -> None:
        shim.check(self.m_device._lib.pt_event_wait(self._h))

    def isComplete(self) -> bool:
        r = self.m_device._lib.pt_event_is_complete(self._h)
        if r < 0:
            shim.check(shim.PT_ERR_HIP)
        return bool(r)

    def waitOnStream(self, hip_stream: int) -> None:
        """Work enqueued later on ``hip_stream`` starts after this event (device-side wait; no Adl counterpart)."""
        shim.check(self.m_device._lib.pt_event_wait_on(self._h, ctypes.c_void_p(int(hip_stream) or shim.PT_STREAM_LEGACY)))

    def getExecutionTimeNanoseconds(self) -> int:
        ns = ctypes.c_uint64()
        shim.check(self.m_device._lib.pt_event_elapsed_ns(self._h, ctypes.byref(ns)))
        return ns.value

    def release(self) -> None:
        if self._h:
            self.m_device._lib.pt_event_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def _ev(sync: Optional[SyncObject]):
    return sync._h if sync is not None else None


class Buffer:
    """adl::Buffer<T> (Adl/Adl.h:203-265).  ``dtype`` plays the role of T; sizes are in elements."""

    def __init__(self, device: Optional[Device] = None, nElems: int = 0, dtype=np.uint8):
        self.m_device = None
        self.m_size = 0
        self._h = None
        self._dtype = np.dtype(dtype)
        self._mapped = None
        if device is not None:
            self.allocate(device, nElems)

    # Buffer<T>::allocate: on failure m_size = 0 and m_ptr = 0 + a log line (AdlCL.inl:190-197)
    def allocate(self, device: Device, nElems: int) -> None:
        self.release()
        self.m_device = device
        out = ctypes.c_void_p()
        rc = device._lib.pt_buffer_alloc(device._h, int(nElems) * self._dtype.itemsize, ctypes.byref(out))
        if rc == shim.PT_ERR_OOM:
            print("HIP Memory Allocation Failure: %s" % device._lib.pt_last_error().decode())
            self.m_size = 0
            self._h = None
            return
        shim.check(rc)
        self._h = out.value
        self.m_size = int(nElems)

    def setRawPtr(self, device: Device, device_ptr: int, nElems: int) -> None:
        """Buffer<T>::setRawPtr (Adl.h:214): adopt caller-owned device memory."""
        self.release()
        self.m_device = device
        out = ctypes.c_void_p()
        shim.check(device._lib.pt_buffer_wrap(device._h, ctypes.c_void_p(device_ptr),
                                              int(nElems) * self._dtype.itemsize, ctypes.byref(out)))
        self._h = out.value
        self.m_size = int(nElems)

    @property
    def m_ptr(self) -> int:
        """The device address as a value (the reference's m_ptr holds the opaque cl_mem).  It does not license
        access to the memory behind the API: take the pointer from getInternalObject() for that."""
        return (self.m_device._lib.pt_buffer_address(self._h) or 0) if self._h else 0

    def getInternalObject(self) -> int:
        """Buffer<T>::getInternalObject: the device pointer for code that touches the memory itself (a torch
        tensor view, another library).  Submits deferred frames and ends frame batching for this buffer."""
        return (self.m_device._lib.pt_buffer_device_ptr(self._h) or 0) if self._h else 0

    def getSize(self) -> int:
        return self.m_size

    def _bytes(self, n: int) -> int:
        return int(n) * self._dtype.itemsize

    def write(self, src, nElems: int, dstOffsetNElems: int = 0, syncObj: Optional[SyncObject] = None) -> None:
        """Buffer<T>::write(host ptr | Buffer)  (Adl.h:218,222)."""
        lib = self.m_device._lib
        if isinstance(src, Buffer):
            shim.check(lib.pt_buffer_copy(self._h, src._h, self._bytes(nElems), self._bytes(dstOffsetNElems), 0, _ev(syncObj)))
            return
        a = np.ascontiguousarray(src)
        if a.nbytes < self._bytes(nElems):
            raise ValueError("host source smaller than nElems")
        self._keep = a  # asynchronous copy: keep the source alive until the next sync
        shim.check(lib.pt_buffer_write(self._h, a.ctypes.data_as(ctypes.c_void_p), self._bytes(nElems),
                                       self._bytes(dstOffsetNElems), _ev(syncObj)))

    def read(self, dst, nElems: int, srcOffsetNElems: int = 0, syncObj: Optional[SyncObject] = None) -> None:
        """Buffer<T>::read(host ptr | Buffer)  (Adl.h:220,224)."""
        lib = self.m_device._lib
        if isinstance(dst, Buffer):
            shim.check(lib.pt_buffer_copy(dst._h, self._h, self._bytes(nElems), 0, self._bytes(srcOffsetNElems), _ev(syncObj)))
            return
        if not (isinstance(dst, np.ndarray) and dst.flags.c_contiguous and dst.flags.writeable):
            raise ValueError("read() needs a writable C-contiguous numpy array")
        if dst.nbytes < self._bytes(nElems):
            raise ValueError("host destination smaller than nElems")
        shim.check(lib.pt_buffer_read(self._h, dst.ctypes.data_as(ctypes.c_void_p), self._bytes(nElems),
                                      self._bytes(srcOffsetNElems), _ev(syncObj)))

    def getHostPtr(self, size: int = -1, blocking: bool = False) -> np.ndarray:
        """Buffer<T>::getHostPtr (Adl/Adl.inl:247-252): non-blocking by default -- call
        DeviceUtils.waitForCompletion before touching the returned array, as the reference does
        (test/RaytraceTest.cpp:225-228)."""
        n = self.m_size if size < 0 else int(size)
        p = self.m_device._lib.pt_buffer_map(self._h, self._bytes(n), 1 if blocking else 0)
        if not p:
            shim.check(shim.PT_ERR_INVALID)
        raw = (ctypes.c_ubyte * self._bytes(n)).from_address(p)
        arr = np.frombuffer(raw, dtype=self._dtype.base, count=self._bytes(n) // self._dtype.base.itemsize)
        if self._dtype.shape:
            arr = arr.reshape((n,) + self._dtype.shape)
        elif self._dtype.names:
            arr = np.frombuffer(raw, dtype=self._dtype, count=n)
        self._mapped = p
        return arr

    def returnHostPtr(self, ptr=None) -> None:
        """Buffer<T>::returnHostPtr (Adl/Adl.inl:254-259)."""
        if self._mapped is None:
            raise ValueError("buffer is not mapped")
        shim.check(self.m_device._lib.pt_buffer_unmap(self._h, ctypes.c_void_p(self._mapped)))
        self._mapped = None

    def release(self) -> None:
        if self._h and self.m_device is not None and self.m_device._h:
            self.m_device._lib.pt_buffer_free(self._h)
        self._h = None
        self.m_size = 0
        self._mapped = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class BufferInfo:
    """adl::BufferInfo (Adl/AdlKernel.h:59-69)."""

    def __init__(self, buff: Buffer, isReadOnly: bool = False):
        self.m_buffer = buff
        self.m_isReadOnly = isReadOnly


class Launcher:
    """adl::Launcher (Adl/AdlKernel.h:121-202, Adl/AdlKernel.inl:144-196): positional arguments,
    buffers as handles, constants by value (<= MAX_ARG_SIZE bytes), index auto-incremented."""

    MAX_ARG_SIZE = shim.PT_MAX_ARG_SIZE
    MAX_ARG_COUNT = shim.PT_MAX_ARG_COUNT

    def __init__(self, device: Device, kernel: Optional[Kernel]):
        self.m_deviceData = device
        self.m_kernel = kernel
        self.m_idx = 0
        self._args = (shim.LaunchArg * shim.PT_MAX_ARG_COUNT)()

    def setBuffers(self, buffInfo: Sequence[BufferInfo], n: Optional[int] = None) -> None:
        n = len(buffInfo) if n is None else n
        for i in range(n):
            if self.m_idx >= self.MAX_ARG_COUNT:
                raise ValueError("too many kernel arguments")
            a = self._args[self.m_idx]
            a.is_buffer = 1
            a.read_only = 1 if buffInfo[i].m_isReadOnly else 0
            a.size = 0
            a.buffer = buffInfo[i].m_buffer._h
            self.m_idx += 1

    def setConst(self, consts) -> None:
        """Launcher::setConst<T>(const T&) (Adl/AdlKernel.inl:163-167): by-value bytes."""
        raw = consts if isinstance(consts, (bytes, bytearray)) else np.ascontiguousarray(consts).tobytes()
        if len(raw) > self.MAX_ARG_SIZE:
            raise ValueError("constant larger than MAX_ARG_SIZE")
        if self.m_idx >= self.MAX_ARG_COUNT:
            raise ValueError("too many kernel arguments")
        a = self._args[self.m_idx]
        a.is_buffer = 0
        a.read_only = 0
        a.size = len(raw)
        a.buffer = None
        ctypes.memmove(a.data, raw, len(raw))
        self.m_idx += 1

    def launch1D(self, numThreads: int, localSize: int = ADL_DEFAULT_LOCAL_SIZE_1D,
                 syncObj: Optional[SyncObject] = None) -> float:
        return self.launch2D(numThreads, 1, localSize, 1, syncObj)

    def launch2D(self, numThreadsX: int, numThreadsY: int, localSizeX: int = ADL_DEFAULT_LOCAL_SIZE_2D,
                 localSizeY: int = ADL_DEFAULT_LOCAL_SIZE_2D, syncObj: Optional[SyncObject] = None) -> float:
        ms = ctypes.c_float(0.0)
        kh = self.m_kernel.m_kernel if self.m_kernel is not None else None
        shim.check(self.m_deviceData._lib.pt_launch_2d(self.m_deviceData._h, kh, self._args, self.m_idx, numThreadsX,
                                                       numThreadsY, localSizeX, localSizeY, _ev(syncObj),
                                                       ctypes.byref(ms)))
        return ms.value
