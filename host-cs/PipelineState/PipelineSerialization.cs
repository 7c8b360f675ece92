// PipelineSerialization.cs -- tile persistence in the reference's on-disk format, Unity-free:
//   <base>/save__<alias>/files.json          {"alias":..,"version":..,"files":[{"id":..,"type":..,"size":..},..]}
//                                            (FileDirectory / FileObject, Pipeline/PipelineState/PipelineSerialization.cs:15-97;
//                                            JsonUtility.ToJson: compact, fields in declaration order)
//   <base>/save__<alias>/data/<name>.data    the buffer's raw little-endian bytes (BinaryIO.WriteBytes :130-146, GetFQN :206-208)
// `type` is typeof(T).Name of the CONTAINER the state manager was asked for (PipelineStateManager.cs:64,111), i.e. the CLR name
// "NativeArray`1" for every float plane of this path; `size` is the element count.  A plane written here is found by the
// reference's PipelineStateManager.GetBuffer and vice versa.  Device tiles cross PCIe through nz_tile_download / nz_tile_upload
// (DeviceTile.ToArray / CopyFrom).  Same behaviour as noize_job_amd/persistence.py, which the test suite drives.
// Source only (no .NET toolchain in the build image).
using System;
using System.Collections.Generic;
using System.IO;
using System.Text;

namespace xshazwar.noize.hip {

    public struct FileObject {                   // PipelineSerialization.cs:91-97
        public string id;
        public string type;
        public int size;
    }

    public sealed class FileDirectory {          // :15-89
        public string alias;
        public string version;
        public List<FileObject> files = new List<FileObject>();
        readonly Dictionary<string, int> lookup = new Dictionary<string, int>();
        public string fullPath;

        public static FileDirectory FromFile(string basePath, string alias = "", string version = "") {    // :24-43
            FileDirectory fd = new FileDirectory { alias = alias, version = version,
                                                   fullPath = Path.Combine(basePath, $"save__{alias}", "files.json") };
            if (File.Exists(fd.fullPath)) fd.Parse(File.ReadAllText(fd.fullPath));
            fd.Init();
            return fd;
        }

        void Init() {                                                                                         // :45-53
            lookup.Clear();
            for (int i = 0; i < files.Count; i++) lookup[$"{files[i].id}_{files[i].type}"] = i;
        }

        public int GetCount(string name, string type) =>                                                      // :55-64
            lookup.TryGetValue($"{name}_{type}", out int idx) ? files[idx].size : -1;

        public void SetCount(string name, string type, int size) {                                            // :71-89
            string key = $"{name}_{type}";
            if (lookup.TryGetValue(key, out int idx)) {
                FileObject f = files[idx];
                f.size = size;
                files[idx] = f;
            } else {
                files.Add(new FileObject { id = name, type = type, size = size });
                lookup[key] = files.Count - 1;
            }
            FlushToDisk();
        }

        public void FlushToDisk() {                                                                           // :66-69
            Directory.CreateDirectory(Path.GetDirectoryName(fullPath));
            File.WriteAllText(fullPath, ToJson());
        }

        // JsonUtility.ToJson(this): compact, declaration order
        public string ToJson() {
            StringBuilder sb = new StringBuilder();
            sb.Append("{\"alias\":").Append(Quote(alias)).Append(",\"version\":").Append(Quote(version)).Append(",\"files\":[");
            for (int i = 0; i < files.Count; i++) {
                if (i > 0) sb.Append(',');
                sb.Append("{\"id\":").Append(Quote(files[i].id)).Append(",\"type\":").Append(Quote(files[i].type))
                  .Append(",\"size\":").Append(files[i].size).Append('}');
            }
            return sb.Append("]}").ToString();
        }

        static string Quote(string s) {
            StringBuilder sb = new StringBuilder("\"");
            foreach (char c in s ?? "") {
                if (c == '"' || c == '\\') sb.Append('\\').Append(c);
                else if (c < 0x20) sb.Append("\\u").Append(((int) c).ToString("x4"));
                else sb.Append(c);
            }
            return sb.Append('"').ToString();
        }

        // the three fields of the file above, in any order and spacing (a hand-rolled reader keeps the host free of a JSON
        // package).  Like the reference's JsonUtility.FromJson, fields it does not know are skipped, not refused.
        void Parse(string text) {
            int i = 0;
            files.Clear();
            Expect(text, ref i, '{');
            while (true) {
                string key = ReadString(text, ref i);
                Expect(text, ref i, ':');
                if (key == "alias") alias = ReadString(text, ref i);
                else if (key == "version") version = ReadString(text, ref i);
                else if (key == "files") {
                    Expect(text, ref i, '[');
                    if (Peek(text, ref i) == ']') { i++; }
                    else {
                        while (true) {
                            FileObject f = new FileObject();
                            Expect(text, ref i, '{');
                            while (true) {
                                string k = ReadString(text, ref i);
                                Expect(text, ref i, ':');
                                if (k == "id") f.id = ReadString(text, ref i);
                                else if (k == "type") f.type = ReadString(text, ref i);
                                else if (k == "size") f.size = ReadInt(text, ref i);
                                else SkipValue(text, ref i);
                                if (Peek(text, ref i) == ',') { i++; continue; }
                                Expect(text, ref i, '}');
                                break;
                            }
                            files.Add(f);
                            if (Peek(text, ref i) == ',') { i++; continue; }
                            Expect(text, ref i, ']');
                            break;
                        }
                    }
                } else SkipValue(text, ref i);
                if (Peek(text, ref i) == ',') { i++; continue; }
                Expect(text, ref i, '}');
                break;
            }
        }
        static char Peek(string t, ref int i) { while (i < t.Length && char.IsWhiteSpace(t[i])) i++; return i < t.Length ? t[i] : '\0'; }
        static void Expect(string t, ref int i, char c) { if (Peek(t, ref i) != c) throw new FormatException($"files.json: '{c}' expected at {i}"); i++; }
        static int ReadInt(string t, ref int i) {
            Peek(t, ref i);
            int s = i;
            while (i < t.Length && (char.IsDigit(t[i]) || t[i] == '-')) i++;
            return int.Parse(t.Substring(s, i - s));
        }
        // any JSON value, read and discarded: string, number / true / false / null, object, array (nested as deep as it goes)
        static void SkipValue(string t, ref int i) {
            char c = Peek(t, ref i);
            if (c == '"') { ReadString(t, ref i); return; }
            if (c == '{' || c == '[') {
                char close = c == '{' ? '}' : ']';
                i++;
                if (Peek(t, ref i) == close) { i++; return; }
                while (true) {
                    if (c == '{') { ReadString(t, ref i); Expect(t, ref i, ':'); }
                    SkipValue(t, ref i);
                    if (Peek(t, ref i) == ',') { i++; continue; }
                    Expect(t, ref i, close);
                    return;
                }
            }
            int s = i;
            while (i < t.Length && t[i] != ',' && t[i] != '}' && t[i] != ']' && !char.IsWhiteSpace(t[i])) i++;
            if (i == s) throw new FormatException($"files.json: a value expected at {i}");
        }
        static string ReadString(string t, ref int i) {
            Expect(t, ref i, '"');
            StringBuilder sb = new StringBuilder();
            while (t[i] != '"') {
                if (t[i] == '\\') {
                    i++;
                    if (t[i] == 'u') { sb.Append((char) Convert.ToInt32(t.Substring(i + 1, 4), 16)); i += 4; }
                    else {
                        char e = t[i];   // \" \\ \/ stand for themselves
                        sb.Append(e == 'n' ? '\n' : e == 't' ? '\t' : e == 'r' ? '\r' : e == 'b' ? '\b' : e == 'f' ? '\f' : e);
                    }
                } else sb.Append(t[i]);
                i++;
            }
            i++;
            return sb.ToString();
        }
    }

    public sealed class PipelineSerdeManager {   // :184-236
        public const string NATIVE_ARRAY = "NativeArray`1", NATIVE_LIST = "NativeList`1", NATIVE_REFERENCE = "NativeReference`1";
        string basePath;
        readonly string alias, version;
        readonly FileDirectory directory;

        public PipelineSerdeManager(string path, string alias, string version) {                             // :192-198
            this.alias = alias;
            this.version = version;
            SetPath(path);
            directory = FileDirectory.FromFile(path, alias, version);
        }
        public void SetPath(string path) { basePath = path; }

        static string CleanFileName(string name) {                                                            // :201-204
            char[] invalids = Path.GetInvalidFileNameChars();
            return string.Join("_", name.Split(invalids, StringSplitOptions.RemoveEmptyEntries)).TrimEnd('.');
        }
        public string GetFQN(string name) => Path.Combine(basePath, $"save__{alias}", "data", $"{CleanFileName(name)}.data");  // :206-208

        // WriteData<T> :210-215 with T = NativeArray<float>: the plane's raw bytes, then the index entry
        public void WriteData(float[] data, int size, string name, string container = NATIVE_ARRAY) {
            string path = GetFQN(name);
            Directory.CreateDirectory(Path.GetDirectoryName(path));
            byte[] bytes = new byte[size * sizeof(float)];
            Buffer.BlockCopy(data, 0, bytes, 0, bytes.Length);   // little-endian hosts only, as the reference's MemCpy
            File.WriteAllBytes(path, bytes);
            directory.SetCount(name, container, size);
        }

        // ReadData :217-224: false when there is no file ("No current file for {name}")
        public bool ReadData(float[] target, int count, string name) {
            string path = GetFQN(name);
            if (!File.Exists(path)) return false;
            byte[] bytes = File.ReadAllBytes(path);
            Buffer.BlockCopy(bytes, 0, target, 0, Math.Min(bytes.Length, Math.Min(count, target.Length) * sizeof(float)));
            return true;
        }

        public int CachedSize(string name, string container = NATIVE_ARRAY) => directory.GetCount(name, container);   // :226-231
    }
}
