"""Tile persistence in the reference's on-disk format, host side only.

Pipeline/PipelineState/PipelineSerialization.cs:
  <base>/save__<alias>/files.json          {"alias":..,"version":..,"files":[{"id":..,"type":..,"size":..},..]}
                                           (FileDirectory / FileObject :15-97, JsonUtility field order)
  <base>/save__<alias>/data/<name>.data    the buffer's raw little-endian bytes (BinaryIO.WriteBytes :130-146,
                                           PipelineSerdeManager.GetFQN :206-208)
`type` is `typeof(T).Name` (:218,231) where T is the CONTAINER type the state manager was asked for
(PipelineStateManager.cs:64,111: SaveBufferToDisk<V, T> / GetBuffer<V, T> with T = NativeArray<float>), i.e. the CLR
name "NativeArray`1" for every plane of this path -- the element type is not recorded; `size` is the element count.
A plane written here is found by the reference's PipelineStateManager.GetBuffer and vice versa.  (Directories written
by this package's first version carried the element name, "Single"; they are still found on read.)  Device tiles go
through nz_tile_download / nz_tile_upload.
"""
import json
import os

import numpy as np

# typeof(T).Name of the serialised container types (ConstraintsLinear.SERIALIZED_TYPES, PipelineState.cs:53-57)
NATIVE_ARRAY, NATIVE_LIST, NATIVE_REFERENCE = "NativeArray`1", "NativeList`1", "NativeReference`1"
# element names this package's first version wrote into `type`; accepted when reading
_LEGACY_ELEMENT_NAMES = {np.dtype(np.float32): "Single", np.dtype(np.int32): "Int32", np.dtype(np.uint32): "UInt32",
                         np.dtype(np.float64): "Double", np.dtype(np.uint8): "Byte", np.dtype(np.int16): "Int16",
                         np.dtype(np.uint16): "UInt16"}
_INVALID = "/\0"  # System.IO.Path.GetInvalidFileNameChars() on Unix


def clean_file_name(name):
    """PipelineSerdeManager.CleanFileName :201-204: split on invalid characters (dropping empty pieces),
    join with '_', trim trailing dots."""
    parts, cur = [], ""
    for ch in name:
        if ch in _INVALID:
            if cur:
                parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur:
        parts.append(cur)
    return "_".join(parts).rstrip(".")


class FileDirectory:  # :15-89
    def __init__(self, base_path, alias="", version=""):
        self.fullPath = os.path.join(base_path, "save__%s" % alias, "files.json")
        if os.path.exists(self.fullPath):
            with open(self.fullPath) as f:
                d = json.load(f)
            self.alias, self.version, self.files = d.get("alias", alias), d.get("version", version), d.get("files", [])
        else:
            self.alias, self.version, self.files = alias, version, []
        self.lookup = {"%s_%s" % (f["id"], f["type"]): i for i, f in enumerate(self.files)}

    def GetCount(self, name, type_):
        i = self.lookup.get("%s_%s" % (name, type_))
        return self.files[i]["size"] if i is not None else -1

    def SetCount(self, name, type_, size):
        key = "%s_%s" % (name, type_)
        if key in self.lookup:
            self.files[self.lookup[key]]["size"] = int(size)
        else:
            self.files.append({"id": name, "type": type_, "size": int(size)})
            self.lookup[key] = len(self.files) - 1
        self.FlushToDisk()

    def FlushToDisk(self):
        os.makedirs(os.path.dirname(self.fullPath), exist_ok=True)
        with open(self.fullPath, "w") as f:  # JsonUtility.ToJson: compact, declaration order
            json.dump({"alias": self.alias, "version": self.version, "files": self.files}, f, separators=(",", ":"))


class PipelineSerdeManager:  # :184-236
    def __init__(self, path, alias, version):
        self.basePath, self.alias, self.version = path, alias, version
        self.directory = FileDirectory(path, alias, version)

    def GetFQN(self, name):
        return os.path.join(self.basePath, "save__%s" % self.alias, "data", "%s.data" % clean_file_name(name))

    def WriteData(self, data, name, container=NATIVE_ARRAY):
        """WriteData<T> :214-219.  data: numpy array or DeviceTile; container = typeof(T).Name."""
        host = data.ToArray() if hasattr(data, "ToArray") else np.ascontiguousarray(data)
        path = self.GetFQN(name)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        host.reshape(-1).astype(host.dtype.newbyteorder("<"), copy=False).tofile(path)
        self.directory.SetCount(name, container, host.size)

    def CachedSize(self, name, dtype=np.float32, container=NATIVE_ARRAY):
        """CachedSize<T> :230-234; -1 when the index has no entry."""
        r = self.directory.GetCount(name, container)
        if r < 0 and np.dtype(dtype) in _LEGACY_ELEMENT_NAMES:
            r = self.directory.GetCount(name, _LEGACY_ELEMENT_NAMES[np.dtype(dtype)])
        return r

    def ReadData(self, name, target=None, dtype=np.float32):
        """Returns the stored array, or fills `target` (numpy array or DeviceTile); None when no file exists
        ("No current file for {name}", :221-224)."""
        path = self.GetFQN(name)
        if not os.path.exists(path):
            return None
        host = np.fromfile(path, dtype=np.dtype(dtype).newbyteorder("<")).astype(dtype, copy=False)
        if target is None:
            return host
        if hasattr(target, "CopyFrom"):
            target.CopyFrom(host[:target.Length])
        else:
            target.reshape(-1)[:host.size] = host[:target.size]
        return target
