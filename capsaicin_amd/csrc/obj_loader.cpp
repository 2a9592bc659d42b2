// obj_loader.cpp — OBJ/MTL parser behind the tinyobj-shaped surface of obj_loader.h, plus the C ABI of
// include/capsaicin_scene.h: LoadObjFile's per-shape (vertex, normal, texcoord) de-duplication and the CPU half of
// CreateGeometryStorage (reference src/systems/asset_load_system.cpp:43-160, 162-233).
#include "obj_loader.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <thread>
#include <tuple>

#include "../../include/capsaicin_scene.h"

namespace tinyobj
{
namespace
{
// Tokenise on blanks/tabs; strips a trailing '\r'.
std::vector<std::string> split_ws(const std::string& line)
{
    std::vector<std::string> out;
    size_t                   i = 0, n = line.size();
    while (i < n)
    {
        while (i < n && (line[i] == ' ' || line[i] == '\t' || line[i] == '\r')) ++i;
        size_t j = i;
        while (j < n && line[j] != ' ' && line[j] != '\t' && line[j] != '\r') ++j;
        if (j > i) out.emplace_back(line, i, j - i);
        i = j;
    }
    return out;
}

bool parse_float(const std::string& s, float* out)
{
    char*  end = nullptr;
    double v   = std::strtod(s.c_str(), &end);
    if (end == s.c_str()) return false;
    *out = (float)v;
    return true;
}

// (OBJ indices are 1-based; negative values count back from the current end of the attribute array: fast_index below)
std::string join_from(const std::vector<std::string>& t, size_t first)
{
    std::string s;
    for (size_t i = first; i < t.size(); ++i) s += (i > first ? " " : "") + t[i];
    return s;
}

bool load_mtl(const std::string& path, std::vector<material_t>* materials, std::map<std::string, int>* by_name)
{
    std::ifstream f(path);
    if (!f) return false;
    std::string line;
    material_t* cur = nullptr;
    while (std::getline(f, line))
    {
        auto t = split_ws(line);
        if (t.empty() || t[0][0] == '#') continue;
        if (t[0] == "newmtl")
        {
            materials->emplace_back();
            cur             = &materials->back();
            cur->name       = join_from(t, 1);
            (*by_name)[cur->name] = (int)materials->size() - 1;
        }
        else if (cur && (t[0] == "Kd" || t[0] == "Ks" || t[0] == "Ke") && t.size() >= 4)
        {
            float* dst = t[0] == "Kd" ? cur->diffuse : t[0] == "Ks" ? cur->specular : cur->emission;
            for (int k = 0; k < 3; ++k) parse_float(t[1 + k], &dst[k]);
        }
        else if (cur && t[0] == "Ns" && t.size() >= 2)
            parse_float(t[1], &cur->shininess);
        else if (cur && t[0] == "map_Kd" && t.size() >= 2)
            cur->diffuse_texname = t.back();
    }
    return true;
}
}  // namespace

// ---- the hot records (v / vn / vt / f: all but a few dozen lines of a scene file) are parsed in place, round 6 ----
// (std::getline + a std::string per token + strtod per number parsed the 262 k-triangle hall's 26 MB at 170 MB/s: bench.py `ingest`)
namespace
{
const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                           1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

inline bool is_blank(char c) { return c == ' ' || c == '\t' || c == '\r'; }

// (float)strtod(token) for the token [p, e): decimal literals of at most 15 significant digits and a power of ten within 1e+-22 are
// converted exactly like strtod does (an exact integer times or over an exact power of ten: one correctly rounded operation, W. D.
// Clinger 1990); anything else -- longer mantissas, inf / nan / hex, a bare sign -- goes to strtod itself.  Like parse_float(): true when
// a number starts the token; what follows it inside the token is ignored.
bool fast_float(const char* p, const char* e, float* out)
{
    const char* q   = p;
    bool        neg = false;
    if (q < e && (*q == '-' || *q == '+')) neg = *q == '-', ++q;
    uint64_t mant = 0;
    int      digits = 0, exp10 = 0;
    bool     any = false, simple = true;
    for (; q < e && *q >= '0' && *q <= '9'; ++q)
    {
        any = true;
        if (mant == 0 && *q == '0') continue;
        if (digits < 19) mant = mant * 10 + (uint64_t)(*q - '0'), ++digits;
        else ++exp10, simple = false;
    }
    if (q < e && *q == '.')
    {
        ++q;
        for (; q < e && *q >= '0' && *q <= '9'; ++q)
        {
            any = true;
            if (mant == 0 && *q == '0')
            {
                --exp10;
                continue;
            }
            if (digits < 19) mant = mant * 10 + (uint64_t)(*q - '0'), ++digits, --exp10;
            else simple = false;
        }
    }
    if (any && q < e && (*q == 'e' || *q == 'E'))
    {
        const char* r    = q + 1;
        bool        eneg = false;
        if (r < e && (*r == '-' || *r == '+')) eneg = *r == '-', ++r;
        if (r < e && *r >= '0' && *r <= '9')
        {
            int ex = 0;
            for (; r < e && *r >= '0' && *r <= '9'; ++r) ex = ex < 10000 ? ex * 10 + (*r - '0') : ex;
            exp10 += eneg ? -ex : ex;
            q = r;
        }
    }
    if (any && simple && q == e && digits <= 15 && exp10 >= -22 && exp10 <= 22)  // (q == e: nothing behind the number inside the token)
    {
        double d = (double)mant;
        d        = exp10 < 0 ? d / kPow10[-exp10] : d * kPow10[exp10];
        *out     = (float)(neg ? -d : d);
        return true;
    }
    // the general case, on a terminated copy of the token
    char        buf[64];
    std::string big;
    const size_t n = (size_t)(e - p);
    const char*  z;
    if (n < sizeof(buf))
        std::memcpy(buf, p, n), buf[n] = 0, z = buf;
    else
        big.assign(p, e), z = big.c_str();
    char*  endp = nullptr;
    double v    = std::strtod(z, &endp);
    if (endp == z) return false;
    *out = (float)v;
    return true;
}

// fix_index() on [p, e)
bool fast_index(const char* p, const char* e, int count, int* out)
{
    const char* q   = p;
    bool        neg = false;
    if (q < e && (*q == '-' || *q == '+')) neg = *q == '-', ++q;
    if (!(q < e && *q >= '0' && *q <= '9')) return false;
    long v = 0;
    for (; q < e && *q >= '0' && *q <= '9'; ++q) v = v < (1L << 40) ? v * 10 + (*q - '0') : v;
    if (neg) v = -v;
    if (v == 0) return false;
    *out = v > 0 ? (int)v - 1 : count + (int)v;
    return *out >= 0 && *out < count;
}
}  // namespace

// ---- the hot records on several threads (round 6; bench.py `ingest`) ----
// The parse below is one pass over the text whose cost is the numbers: 2 M floats and 2.4 M indices in the 262 k-triangle hall.  For a
// large file the text is cut at line ends into one chunk per thread; pass A converts a chunk's v / vn / vt records into chunk-local
// arrays and notes its f records (with the chunk-local attribute counts at that line: an index is valid against what the file has
// defined SO FAR) and its rare records; pass B, once the counts in front of every chunk are known, converts the f records.  The merge
// walks the chunks in file order and does what the sequential loop does with the rare records.  ANY anomaly -- a malformed number, an
// index out of range, a short face -- abandons the attempt: the sequential parser then runs from the start and reports it, so error
// texts, line numbers and the accept / reject decisions are its alone, and for a file without one the arrays are the same bytes.
int g_obj_threads = 0;  // cap_obj_set_threads: 0 = one per hardware thread (at most 8) for files above 1 MB, 1 = the sequential parser only

struct ObjChunk
{
    const char *begin = nullptr, *end = nullptr;
    std::vector<float> v, vn, vt;
    struct Face
    {
        const char *a, *e;         // the record's text behind the keyword
        uint32_t    nv, nvt, nvn;  // chunk-local attribute counts in front of this line
    };
    struct Event
    {
        uint32_t    face_pos;  // faces of this chunk in front of the record
        const char *k0, *eol;
    };
    std::vector<Face>     faces;
    std::vector<Event>    events;
    std::vector<index_t>  tris;       // pass B: three per triangle
    std::vector<uint32_t> tri_start;  // pass B: first triangle of face i (size faces + 1)
    uint32_t base_v = 0, base_vt = 0, base_vn = 0;  // attribute counts in front of the chunk
    bool     anomaly = false;
};

void obj_chunk_pass_a(ObjChunk& c)
{
    const char* p = c.begin;
    while (p < c.end)
    {
        const char* eol = (const char*)std::memchr(p, '\n', (size_t)(c.end - p));
        if (!eol) eol = c.end;
        const char* q = p;
        p             = eol < c.end ? eol + 1 : c.end;
        auto token = [&](const char*& a, const char*& b) {
            while (q < eol && is_blank(*q)) ++q;
            a = q;
            while (q < eol && !is_blank(*q)) ++q;
            b = q;
            return b > a;
        };
        const char *k0, *k1;
        if (!token(k0, k1) || *k0 == '#') continue;
        const size_t kl = (size_t)(k1 - k0);
        const bool   is_v = kl == 1 && k0[0] == 'v', is_vn = kl == 2 && k0[0] == 'v' && k0[1] == 'n';
        if (is_v || is_vn)
        {
            float       v[3];
            const char *a[3], *b[3];
            for (int k = 0; k < 3; ++k)
                if (!token(a[k], b[k])) { c.anomaly = true; return; }
            for (int k = 0; k < 3; ++k)
                if (!fast_float(a[k], b[k], &v[k])) { c.anomaly = true; return; }
            auto& dst = is_v ? c.v : c.vn;
            dst.insert(dst.end(), v, v + 3);
        }
        else if (kl == 2 && k0[0] == 'v' && k0[1] == 't')
        {
            const char *a, *b;
            if (!token(a, b)) { c.anomaly = true; return; }
            float u = 0.f, v = 0.f;
            if (!fast_float(a, b, &u)) { c.anomaly = true; return; }
            if (token(a, b) && !fast_float(a, b, &v)) { c.anomaly = true; return; }
            c.vt.push_back(u), c.vt.push_back(v);
        }
        else if (kl == 1 && k0[0] == 'f')
            c.faces.push_back({q, eol, (uint32_t)(c.v.size() / 3), (uint32_t)(c.vt.size() / 2), (uint32_t)(c.vn.size() / 3)});
        else
            c.events.push_back({(uint32_t)c.faces.size(), k0, eol});
    }
}

void obj_chunk_pass_b(ObjChunk& c)
{
    std::vector<index_t> face;
    c.tri_start.reserve(c.faces.size() + 1);
    c.tris.reserve(c.faces.size() * 6);
    for (const ObjChunk::Face& f : c.faces)
    {
        c.tri_start.push_back((uint32_t)(c.tris.size() / 3));
        const int nv = (int)(c.base_v + f.nv), nvt = (int)(c.base_vt + f.nvt), nvn = (int)(c.base_vn + f.nvn);
        const char* q = f.a;
        face.clear();
        while (true)
        {
            while (q < f.e && is_blank(*q)) ++q;
            const char* a = q;
            while (q < f.e && !is_blank(*q)) ++q;
            const char* b = q;
            if (b == a) break;
            index_t     idx{-1, -1, -1};
            const char* s1 = (const char*)std::memchr(a, '/', (size_t)(b - a));
            const char* s2 = s1 ? (const char*)std::memchr(s1 + 1, '/', (size_t)(b - s1 - 1)) : nullptr;
            const char* ae = s1 ? s1 : b;
            if (!fast_index(a, ae, nv, &idx.vertex_index)) { c.anomaly = true; return; }
            if (s1)
            {
                const char* tb = s1 + 1;
                const char* te = s2 ? s2 : b;
                if (te > tb && !fast_index(tb, te, nvt, &idx.texcoord_index)) { c.anomaly = true; return; }
                if (s2 && b > s2 + 1 && !fast_index(s2 + 1, b, nvn, &idx.normal_index)) { c.anomaly = true; return; }
            }
            face.push_back(idx);
        }
        if (face.size() < 3) { c.anomaly = true; return; }
        for (size_t j = 2; j < face.size(); ++j) c.tris.push_back(face[0]), c.tris.push_back(face[j - 1]), c.tris.push_back(face[j]);
    }
    c.tri_start.push_back((uint32_t)(c.tris.size() / 3));
}

template <class F>
void obj_parallel_for(std::vector<ObjChunk>& chunks, F fn)
{
    std::vector<std::thread> pool;
    for (size_t i = 1; i < chunks.size(); ++i) pool.emplace_back([&chunks, i, fn] { fn(chunks[i]); });
    fn(chunks[0]);
    for (auto& t : pool) t.join();
}

// handle_rare: the sequential loop's treatment of an o / g / usemtl / mtllib record (shared).  false: an anomaly; the caller starts over.
template <class Rare>
bool load_obj_parallel(const std::string& text, int threads, attrib_t* attrib, shape_t& cur, int& cur_material, Rare handle_rare)
{
    std::vector<ObjChunk> chunks((size_t)threads);
    const char* const     base = text.data();
    const size_t          n    = text.size();
    size_t                at   = 0;
    for (int i = 0; i < threads; ++i)
    {
        size_t stop = i + 1 == threads ? n : n * (size_t)(i + 1) / (size_t)threads;
        if (stop < at) stop = at;
        if (stop < n)
        {
            const char* nl = (const char*)std::memchr(base + stop, '\n', n - stop);
            stop           = nl ? (size_t)(nl - base) + 1 : n;
        }
        chunks[(size_t)i].begin = base + at, chunks[(size_t)i].end = base + stop;
        at = stop;
    }
    obj_parallel_for(chunks, obj_chunk_pass_a);
    uint32_t nv = 0, nvt = 0, nvn = 0;
    for (ObjChunk& c : chunks)
    {
        if (c.anomaly) return false;
        c.base_v = nv, c.base_vt = nvt, c.base_vn = nvn;
        nv += (uint32_t)(c.v.size() / 3), nvt += (uint32_t)(c.vt.size() / 2), nvn += (uint32_t)(c.vn.size() / 3);
    }
    obj_parallel_for(chunks, obj_chunk_pass_b);
    for (const ObjChunk& c : chunks)
        if (c.anomaly) return false;
    attrib->vertices.reserve((size_t)nv * 3), attrib->texcoords.reserve((size_t)nvt * 2), attrib->normals.reserve((size_t)nvn * 3);
    for (const ObjChunk& c : chunks)
    {
        attrib->vertices.insert(attrib->vertices.end(), c.v.begin(), c.v.end());
        attrib->normals.insert(attrib->normals.end(), c.vn.begin(), c.vn.end());
        attrib->texcoords.insert(attrib->texcoords.end(), c.vt.begin(), c.vt.end());
        auto append = [&](uint32_t f0, uint32_t f1) {  // faces [f0, f1) of the chunk to the current shape, with the current material
            if (f1 <= f0) return;
            const uint32_t t0 = c.tri_start[f0], t1 = c.tri_start[f1];
            cur.mesh.indices.insert(cur.mesh.indices.end(), c.tris.begin() + 3 * (size_t)t0, c.tris.begin() + 3 * (size_t)t1);
            cur.mesh.material_ids.insert(cur.mesh.material_ids.end(), (size_t)(t1 - t0), cur_material);
        };
        uint32_t pos = 0;
        for (const ObjChunk::Event& e : c.events)
        {
            append(pos, e.face_pos);
            pos = e.face_pos;
            handle_rare(e.k0, e.eol);
        }
        append(pos, (uint32_t)c.faces.size());
    }
    return true;
}

bool LoadObj(attrib_t* attrib, std::vector<shape_t>* shapes, std::vector<material_t>* materials, std::string* warn,
             std::string* err, const char* filename, const char* mtl_basedir)
{
    attrib->vertices.clear(), attrib->normals.clear(), attrib->texcoords.clear();
    shapes->clear(), materials->clear();
    std::string text;
    {
        std::ifstream f(filename, std::ios::binary);
        if (!f)
        {
            if (err) *err += std::string("Cannot open file [") + filename + "]\n";
            return false;
        }
        f.seekg(0, std::ios::end);
        const std::streamoff size = f.tellg();
        f.seekg(0, std::ios::beg);
        if (size > 0)
        {
            text.resize((size_t)size);
            f.read(&text[0], size);
            text.resize((size_t)f.gcount());
        }
    }
    std::string basedir;
    if (mtl_basedir && *mtl_basedir)
    {
        basedir = mtl_basedir;
        if (basedir.back() != '/') basedir += '/';
    }
    else
    {
        std::string fn(filename);
        size_t      slash = fn.find_last_of('/');
        basedir           = slash == std::string::npos ? "" : fn.substr(0, slash + 1);
    }
    std::map<std::string, int> material_by_name;
    shape_t                    cur;
    int                        cur_material = -1;
    size_t                     line_no = 0;
    auto flush = [&]() {
        if (!cur.mesh.indices.empty()) shapes->push_back(cur);
        cur = shape_t();
    };
    // the rare records: o / g / usemtl / mtllib (s, l, p and unknown records are ignored); tokens as strings
    auto handle_rare = [&](const char* k0, const char* eol) {
        const std::string line(k0, eol);
        auto              t = split_ws(line);
        const std::string& k = t[0];
        if (k == "o" || k == "g")
        {
            flush();
            cur.name = join_from(t, 1);
        }
        else if (k == "usemtl")
        {
            auto it      = material_by_name.find(join_from(t, 1));
            cur_material = it == material_by_name.end() ? -1 : it->second;
        }
        else if (k == "mtllib")
        {
            for (size_t i = 1; i < t.size(); ++i)
                if (!load_mtl(basedir + t[i], materials, &material_by_name) && warn)
                    *warn += "Material file [ " + t[i] + " ] not found in a path : " + basedir + "\n";
        }
    };
    {
        int threads = g_obj_threads;
        if (threads <= 0)
        {
            const unsigned hw = std::thread::hardware_concurrency();
            threads           = text.size() < ((size_t)1 << 20) ? 1 : (int)(hw == 0 ? 1 : (hw > 8 ? 8 : hw));
        }
        if (threads > 1)
        {
            const std::string warn_before = warn ? *warn : std::string();
            if (load_obj_parallel(text, threads, attrib, cur, cur_material, handle_rare))
            {
                flush();
                return true;
            }
            // an anomaly somewhere: from the start, one line after the other, so that it is reported as it always was
            attrib->vertices.clear(), attrib->normals.clear(), attrib->texcoords.clear();
            shapes->clear(), materials->clear(), material_by_name.clear();
            if (warn) *warn = warn_before;
            cur          = shape_t();
            cur_material = -1;
        }
    }
    std::vector<index_t> face;
    const char*          p   = text.data();
    const char* const    end = p + text.size();
    while (p < end)
    {
        const char* eol = (const char*)std::memchr(p, '\n', (size_t)(end - p));
        if (!eol) eol = end;
        const char* q = p;
        p             = eol < end ? eol + 1 : end;
        ++line_no;
        auto bad = [&](const char* what) {
            if (err) *err += std::string(filename) + ":" + std::to_string(line_no) + ": " + what + "\n";
            return false;
        };
        // next token of the line: [a, b); false at the end of the line
        auto token = [&](const char*& a, const char*& b) {
            while (q < eol && is_blank(*q)) ++q;
            a = q;
            while (q < eol && !is_blank(*q)) ++q;
            b = q;
            return b > a;
        };
        const char *k0, *k1;
        if (!token(k0, k1) || *k0 == '#') continue;
        const size_t kl = (size_t)(k1 - k0);
        const bool   is_v = kl == 1 && k0[0] == 'v', is_vn = kl == 2 && k0[0] == 'v' && k0[1] == 'n';
        if (is_v || is_vn)
        {
            float       v[3];
            const char *a[3], *b[3];
            for (int c = 0; c < 3; ++c)
                if (!token(a[c], b[c])) return bad("expected 3 coordinates");
            for (int c = 0; c < 3; ++c)
                if (!fast_float(a[c], b[c], &v[c])) return bad("malformed number");
            auto& dst = is_v ? attrib->vertices : attrib->normals;
            dst.insert(dst.end(), v, v + 3);
        }
        else if (kl == 2 && k0[0] == 'v' && k0[1] == 't')
        {
            const char *a, *b;
            if (!token(a, b)) return bad("expected texture coordinates");
            float u = 0.f, v = 0.f;
            if (!fast_float(a, b, &u)) return bad("malformed number");
            if (token(a, b) && !fast_float(a, b, &v)) return bad("malformed number");
            attrib->texcoords.push_back(u), attrib->texcoords.push_back(v);
        }
        else if (kl == 1 && k0[0] == 'f')
        {
            face.clear();
            const char *a, *b;
            while (token(a, b))
            {
                index_t     idx{-1, -1, -1};
                const char* s1 = (const char*)std::memchr(a, '/', (size_t)(b - a));
                const char* s2 = s1 ? (const char*)std::memchr(s1 + 1, '/', (size_t)(b - s1 - 1)) : nullptr;
                const char* ae = s1 ? s1 : b;
                if (!fast_index(a, ae, (int)attrib->vertices.size() / 3, &idx.vertex_index)) return bad("vertex index out of range");
                if (s1)
                {
                    const char* tb = s1 + 1;
                    const char* te = s2 ? s2 : b;
                    if (te > tb && !fast_index(tb, te, (int)attrib->texcoords.size() / 2, &idx.texcoord_index)) return bad("texcoord index out of range");
                    if (s2 && b > s2 + 1 && !fast_index(s2 + 1, b, (int)attrib->normals.size() / 3, &idx.normal_index)) return bad("normal index out of range");
                }
                face.push_back(idx);
            }
            if (face.size() < 3) return bad("face with fewer than 3 vertices");
            for (size_t j = 2; j < face.size(); ++j)  // fan: (0, j-1, j)
            {
                cur.mesh.indices.push_back(face[0]);
                cur.mesh.indices.push_back(face[j - 1]);
                cur.mesh.indices.push_back(face[j]);
                cur.mesh.material_ids.push_back(cur_material);
            }
        }
        else
            handle_rare(k0, eol);
    }
    flush();
    return true;
}
}  // namespace tinyobj

// ---------------------------------------------------------------------------------------------------------
// C ABI (include/capsaicin_scene.h)
// ---------------------------------------------------------------------------------------------------------
struct CapGeometry
{
    std::vector<float>       positions, normals, texcoords;
    std::vector<uint32_t>    indices;
    std::vector<CapMeshDesc> meshes;
    std::vector<std::string> texture_names;
    std::vector<CapMaterial> mesh_materials;
    uint32_t                 material_count = 0;
    std::string              warn;
};

// Both halves of the library report through the same cap_last_error(); the setter lives in context.hip.
extern "C" void cap_set_error_(const char* msg);

extern "C" {

int cap_obj_load(const char* obj_path, const char* mtl_dir, CapGeometry** out)
{
    if (!obj_path || !out)
    {
        cap_set_error_("cap_obj_load: NULL argument");
        return CAP_ERR_INVALID_ARG;
    }
    *out = nullptr;
    tinyobj::attrib_t                attrib;
    std::vector<tinyobj::shape_t>    shapes;
    std::vector<tinyobj::material_t> objmaterials;
    std::string                      warn, err;
    // asset_load_system.cpp:54-67: a non-empty err or a false return throws in the reference
    bool ret = tinyobj::LoadObj(&attrib, &shapes, &objmaterials, &warn, &err, obj_path, mtl_dir);
    if (!err.empty() || !ret)
    {
        cap_set_error_(("AssetLoadSystem: Couldn't load " + std::string(obj_path) + ": " + err).c_str());
        return CAP_ERR_IO;
    }
    CapGeometry* g    = new CapGeometry;
    g->warn           = warn;
    g->material_count = (uint32_t)objmaterials.size();
    // asset_load_system.cpp:76-89: texture index per material (name -> index in first-use order, texture_system.cpp:20-36)
    std::vector<uint32_t> texture_indices;
    for (auto& m : objmaterials)
    {
        if (m.diffuse_texname.empty())
        {
            texture_indices.push_back(~0u);
            continue;
        }
        uint32_t idx = 0;
        for (; idx < g->texture_names.size(); ++idx)
            if (g->texture_names[idx] == m.diffuse_texname) break;
        if (idx == g->texture_names.size()) g->texture_names.push_back(m.diffuse_texname);
        texture_indices.push_back(idx);
    }
    const float kd_default = std::pow(0.75f, 2.2f);  // only a display default for the EXT material table
    uint32_t    vertex_total = 0, index_total = 0;
    for (uint32_t si = 0; si < shapes.size(); ++si)
    {
        // asset_load_system.cpp:100-142
        // (vertex, normal, texcoord) -> the shape's vertex number, first use first: an open-addressing table (a std::map of tuples was a
        // third of the load time of a 262 k-triangle scene)
        const auto&    sh = shapes[si];
        size_t         cap = 16;
        while (cap < 2 * sh.mesh.indices.size()) cap <<= 1;
        std::vector<uint32_t> slot(cap, ~0u);  // index into `keys`
        std::vector<tinyobj::index_t> keys;
        keys.reserve(sh.mesh.indices.size() / 2);
        uint32_t       nverts = 0;
        const uint32_t first_index = index_total;
        g->indices.reserve(g->indices.size() + sh.mesh.indices.size());
        for (const auto& ix : sh.mesh.indices)
        {
            uint64_t h = (uint64_t)(uint32_t)ix.vertex_index * 0x9E3779B97F4A7C15ull ^ (uint64_t)(uint32_t)ix.normal_index * 0xC2B2AE3D27D4EB4Full ^
                         (uint64_t)(uint32_t)ix.texcoord_index * 0x165667B19E3779F9ull;
            size_t   at = (size_t)(h >> 20) & (cap - 1);
            uint32_t found = ~0u;
            for (;; at = (at + 1) & (cap - 1))
            {
                const uint32_t k = slot[at];
                if (k == ~0u) break;
                if (keys[k].vertex_index == ix.vertex_index && keys[k].normal_index == ix.normal_index && keys[k].texcoord_index == ix.texcoord_index)
                {
                    found = k;
                    break;
                }
            }
            if (found != ~0u)
            {
                g->indices.push_back(found);
                continue;
            }
            slot[at] = nverts;
            keys.push_back(ix);
            g->indices.push_back(nverts++);
            for (int k = 0; k < 3; ++k) g->positions.push_back(attrib.vertices[3 * ix.vertex_index + k]);
            for (int k = 0; k < 3; ++k) g->normals.push_back(ix.normal_index != -1 ? attrib.normals[3 * ix.normal_index + k] : 0.f);
            for (int k = 0; k < 2; ++k) g->texcoords.push_back(ix.texcoord_index != -1 ? attrib.texcoords[2 * ix.texcoord_index + k] : 0.f);
        }
        index_total += (uint32_t)sh.mesh.indices.size();
        // asset_load_system.cpp:146-150 and 171-179
        const int   mat = sh.mesh.material_ids.empty() ? -1 : sh.mesh.material_ids[0];
        CapMeshDesc d{};
        d.vertex_count        = nverts;
        d.first_vertex_offset = vertex_total;
        d.index_count         = (uint32_t)sh.mesh.indices.size();
        d.first_index_offset  = first_index;
        d.index               = si;
        d.texture_index       = mat == -1 ? ~0u : texture_indices[mat];
        g->meshes.push_back(d);
        vertex_total += nverts;
        CapMaterial cm{};
        if (mat == -1)
            cm.kd[0] = cm.kd[1] = cm.kd[2] = kd_default, cm.roughness = 1.f;
        else
        {
            const auto& m = objmaterials[mat];
            for (int k = 0; k < 3; ++k) cm.kd[k] = m.diffuse[k], cm.ks[k] = m.specular[k], cm.ke[k] = m.emission[k];
            // Blinn-Phong exponent -> GGX alpha = sqrt(2 / (Ns + 2)); roughness = sqrt(alpha)
            cm.roughness = std::sqrt(std::sqrt(2.0f / (m.shininess + 2.0f)));
        }
        g->mesh_materials.push_back(cm);
    }
    *out = g;
    return CAP_OK;
}

void cap_geometry_free(CapGeometry* g) { delete g; }

void cap_obj_set_threads(int threads) { tinyobj::g_obj_threads = threads < 0 ? 0 : threads; }

int cap_geometry_view(const CapGeometry* g, CapGeometryView* out)
{
    if (!g || !out)
    {
        cap_set_error_("cap_geometry_view: NULL argument");
        return CAP_ERR_INVALID_ARG;
    }
    out->positions      = g->positions.data();
    out->normals        = g->normals.data();
    out->texcoords      = g->texcoords.data();
    out->indices        = g->indices.data();
    out->meshes         = g->meshes.data();
    out->vertex_count   = (uint32_t)(g->positions.size() / 3);
    out->index_count    = (uint32_t)g->indices.size();
    out->mesh_count     = (uint32_t)g->meshes.size();
    out->texture_count  = (uint32_t)g->texture_names.size();
    out->material_count = g->material_count;
    return CAP_OK;
}

const char* cap_geometry_texture_name(const CapGeometry* g, uint32_t i)
{
    return (g && i < g->texture_names.size()) ? g->texture_names[i].c_str() : nullptr;
}

const char* cap_geometry_warning(const CapGeometry* g) { return g ? g->warn.c_str() : ""; }

int cap_geometry_materials(const CapGeometry* g, CapMaterial* out)
{
    if (!g || !out)
    {
        cap_set_error_("cap_geometry_materials: NULL argument");
        return CAP_ERR_INVALID_ARG;
    }
    std::memcpy(out, g->mesh_materials.data(), sizeof(CapMaterial) * g->mesh_materials.size());
    return CAP_OK;
}

int cap_scene_upload_geometry(CapContext* ctx, const CapGeometry* g)
{
    CapGeometryView v;
    int             rc = cap_geometry_view(g, &v);
    if (rc) return rc;
    return cap_scene_upload(ctx, v.positions, v.normals, v.texcoords, v.indices, v.meshes, v.vertex_count, v.index_count, v.mesh_count);
}
}
