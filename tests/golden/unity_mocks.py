"""Recording stand-ins for the slice of UnityEngine that AmbientOcclusion.cs touches while it
builds its command buffers.  TEST INFRASTRUCTURE ONLY: used with oracle/csharp_interp.py to run the
reference's own C# (DoLazyInitialization + RebuildCommandBuffers) and capture what it would record:
the temporary-RT allocations (name, dims, slices, format) and every compute dispatch with its
texture bindings, constants and group counts.  Nothing here computes anything on the path."""
from __future__ import annotations

import numpy as np

from oracle import csharp_interp as CS

F = np.float32


class Any:
    """A permissive Unity object / enum member: any attribute exists, any method can be called.
    Two Any compare equal when their dotted names are equal (enum members)."""

    def __init__(self, name, **attrs):
        object.__setattr__(self, "_name", name)
        for k, v in attrs.items():
            object.__setattr__(self, k, v)

    def __getattr__(self, key):
        if key.startswith("__"):
            raise AttributeError(key)
        return Any(self._name + "." + key)

    def __call__(self, *args):
        return Any(self._name + "()")

    def __eq__(self, other):
        return isinstance(other, Any) and other._name == self._name

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash(self._name)

    def __bool__(self):
        return True

    def __repr__(self):
        return "<%s>" % self._name


class Vector:
    def __init__(self, *c):
        self.c = [F(v) for v in c]
        for name, v in zip("xyzw", self.c):
            setattr(self, name, v)


class Matrix:
    def __init__(self, m00):
        self.m00 = F(m00)

    def __getitem__(self, ij):
        assert tuple(ij) == (0, 0)
        return self.m00


class RenderTargetIdentifier:
    def __init__(self, what):
        self.what = what            # a property name (str), a RenderTexture mock, or a builtin Any


class RenderTexture:
    def __init__(self, width, height, depth, fmt, readwrite):
        self.width, self.height, self.format = int(width), int(height), fmt
        self.volumeDepth = 1

    def Release(self):
        pass

    def Create(self):
        pass


class ComputeShader:
    """FindKernel returns the kernel name; thread-group sizes come from the shader's own text."""

    def __init__(self, name, numthreads):
        self.name, self.numthreads = name, numthreads      # numthreads: kernel -> (x, y, z)

    def FindKernel(self, kernel):
        assert kernel in self.numthreads, kernel
        return kernel

    def GetKernelThreadGroupSizes(self, kernel, x, y, z):
        tx, ty, tz = self.numthreads[kernel]
        x.set(tx)
        y.set(ty)
        z.set(tz)


class CommandBuffer:
    """Records allocations and compute dispatches in order; raster commands are only logged."""

    def __init__(self):
        self.name = ""
        self.Clear()

    def Clear(self):
        self.allocs = {}            # property name -> (w, h, slices, format name)
        self.params = {}            # shader name -> {"tex": {}, "const": {}}
        self.dispatches = []
        self.raster = []

    def _p(self, cs):
        return self.params.setdefault(cs.name, {"tex": {}, "const": {}})

    def GetTemporaryRT(self, name, w, h, depth_bits, filt, fmt, rw, aa, uav):
        self.allocs[name] = (int(w), int(h), 1, fmt._name.split(".")[-1])

    def GetTemporaryRTArray(self, name, w, h, slices, depth_bits, filt, fmt, rw, aa, uav):
        self.allocs[name] = (int(w), int(h), int(slices), fmt._name.split(".")[-1])

    def ReleaseTemporaryRT(self, name):
        pass

    def SetComputeTextureParam(self, cs, kernel, prop, rti):
        self._p(cs)["tex"][prop] = rti.what if isinstance(rti, RenderTargetIdentifier) else rti

    def SetComputeVectorParam(self, cs, prop, vec):
        self._p(cs)["const"][prop] = list(vec.c) + [F(0)] * (4 - len(vec.c))

    def SetComputeFloatParam(self, cs, prop, value):
        self._p(cs)["const"][prop] = [F(value)]

    def SetComputeFloatParams(self, cs, prop, values):
        self._p(cs)["const"][prop] = [F(v) for v in values]     # Unity copies the array at record time

    def DispatchCompute(self, cs, kernel, gx, gy, gz):
        p = self._p(cs)
        self.dispatches.append({"shader": cs.name, "kernel": kernel, "groups": (int(gx), int(gy), int(gz)),
                                "tex": dict(p["tex"]), "const": {k: list(v) for k, v in p["const"].items()}})

    def __getattr__(self, key):     # SetRenderTarget, DrawProcedural, Blit, SetGlobalTexture ...
        if key.startswith("__"):
            raise AttributeError(key)
        return lambda *a: self.raster.append(key)


def run_component(ao_cs_path, numthreads, *, width, height, near, far, proj00, reversed_z, properties,
                  stereo=False, want_interp=False):
    """Instantiate the reference's AmbientOcclusion class from its source, run
    DoLazyInitialization() and RebuildCommandBuffers() against the mocks, return the render
    CommandBuffer mock (allocations + dispatches) and the size/format of the persistent result RT.
    properties: serialized field name -> value (e.g. {"_intensity": 1.1}).
    stereo: single-pass stereo as the component detects it (AO.cs:392-401): camera.stereoEnabled,
    no target texture, one draw per frame; width is then the per-eye pixelWidth.
    want_interp: also return the interpreter and the component instance."""
    classes = CS.load(ao_cs_path)
    camera = Any("camera", pixelWidth=int(width), pixelHeight=int(height), nearClipPlane=F(near),
                 farClipPlane=F(far), projectionMatrix=Matrix(proj00), stereoEnabled=bool(stereo),
                 targetTexture=None, allowHDR=True, actualRenderingPath=Any("RenderingPath.DeferredShading"))
    system_info = Any("SystemInfo", usesReversedZBuffer=bool(reversed_z),
                      graphicsDeviceType=Any("GraphicsDeviceType.Direct3D11"))   # resolved depth: no copy blit
    g = {
        "Mathf": Any("Mathf", Sqrt=CS.mathf_sqrt, Pow=CS.mathf_pow),
        "SystemInfo": system_info,
        "Shader": Any("Shader", PropertyToID=lambda name: name),
        "Application": Any("Application", isPlaying=True),
        "GetComponent": lambda: camera,
        "Vector2": Vector, "Vector4": Vector,
        "RenderTargetIdentifier": RenderTargetIdentifier, "RenderTexture": RenderTexture,
        "CommandBuffer": CommandBuffer, "Material": lambda shader: Any("Material"),
    }
    for enum in ("RenderingPath", "GraphicsDeviceType", "CameraEvent", "BuiltinRenderTextureType",
                 "DepthTextureMode", "RenderTextureFormat", "RenderTextureReadWrite", "FilterMode",
                 "TextureDimension", "HideFlags", "MeshTopology", "Matrix4x4"):
        g[enum] = Any(enum)
    it = CS.Interp(classes, g)
    comp = it.new_instance(classes["AmbientOcclusion"], [])
    for k, v in properties.items():
        assert k in comp.f, k
        comp.f[k] = F(v) if isinstance(comp.f[k], np.float32) else v
    for field, shader in (("_downsample1Compute", "Downsample1"), ("_downsample2Compute", "Downsample2"),
                          ("_renderCompute", "Render"), ("_upsampleCompute", "Upsample")):
        comp.f[field] = ComputeShader(shader, numthreads[shader])
    comp.f["_blitShader"] = Any("BlitShader")
    if stereo:
        assert "_drawCountPerFrame" in comp.f
        comp.f["_drawCountPerFrame"] = 1
    cls = classes["AmbientOcclusion"]
    it.call_method(comp, cls, "DoLazyInitialization", [])
    it.call_method(comp, cls, "RebuildCommandBuffers", [])
    cmd = comp.f["_renderCommand"]
    result_rt = comp.f["_result"].f["_rt"]
    if want_interp:
        return cmd, result_rt, it, comp
    return cmd, result_rt
