// rt_xmlload.cpp -- scene-file loader accepting the reference's XML schema.
//
// Schema and semantics follow LoadScene and its helpers in FIN/xmlload.cpp (FIN = /root/
// reference/RayTracingFinal/RayTracingFinal): element names compared case-insensitively
// (:34-38); <object> nodes with type sphere|plane|obj, their child transforms applied in
// document order AFTER the child objects are loaded (:254-259, 265-291); colours and vectors
// as "default value, overridden per attribute, then multiplied by `value`" (:453-486);
// materials of type blinn (:295-371); lights ambient|direct|point (:375-449); camera (:107-127).
// The reference parses with the vendored TinyXML; this file carries its own small reader for
// the subset those scene files use (elements, attributes, comments, self-closing tags).
#include "rt_scene.h"

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <strings.h>

namespace rt {
namespace {

struct XmlElement {
    std::string name;
    std::vector<std::pair<std::string, std::string>> attrs;
    std::vector<std::unique_ptr<XmlElement>> children;
    const char *Attribute(const char *key) const
    {
        for (auto &a : attrs) if (a.first == key) return a.second.c_str();
        return nullptr;
    }
    // TiXmlElement::QueryDoubleAttribute: leaves *d untouched when the attribute is absent
    void QueryDouble(const char *key, double *d) const
    {
        const char *a = Attribute(key);
        if (!a) return;
        double v;
        if (sscanf(a, "%lf", &v) == 1) *d = v;
    }
    void QueryInt(const char *key, int *i) const
    {
        const char *a = Attribute(key);
        if (!a) return;
        int v;
        if (sscanf(a, "%d", &v) == 1) *i = v;
    }
    const XmlElement *FirstChild(const char *n) const
    {
        for (auto &c : children) if (c->name == n) return c.get();
        return nullptr;
    }
};

class XmlParser {
    const std::string &s;
    size_t i = 0;
    std::string *err;
    bool Fail(const char *m) { if (err && err->empty()) *err = std::string("XML: ") + m + " at byte " + std::to_string(i); return false; }
    void SkipSpace() { while (i < s.size() && isspace((unsigned char)s[i])) i++; }
    bool SkipMisc()
    {
        for (;;) {
            SkipSpace();
            if (s.compare(i, 4, "<!--") == 0) {
                size_t e = s.find("-->", i + 4);
                if (e == std::string::npos) return Fail("unterminated comment");
                i = e + 3;
            } else if (s.compare(i, 2, "<?") == 0) {
                size_t e = s.find("?>", i + 2);
                if (e == std::string::npos) return Fail("unterminated declaration");
                i = e + 2;
            } else if (s.compare(i, 2, "<!") == 0) {
                size_t e = s.find('>', i);
                if (e == std::string::npos) return Fail("unterminated <!");
                i = e + 1;
            } else return true;
        }
    }
    static bool NameChar(char c) { return isalnum((unsigned char)c) || c == '_' || c == '-' || c == ':' || c == '.'; }
public:
    XmlParser(const std::string &text, std::string *e) : s(text), err(e) {}
    std::unique_ptr<XmlElement> ParseElement()
    {
        if (!SkipMisc()) return nullptr;
        if (i >= s.size() || s[i] != '<') { Fail("expected '<'"); return nullptr; }
        i++;
        std::unique_ptr<XmlElement> e(new XmlElement);
        while (i < s.size() && NameChar(s[i])) e->name += s[i++];
        if (e->name.empty()) { Fail("empty element name"); return nullptr; }
        for (;;) {
            SkipSpace();
            if (i >= s.size()) { Fail("unexpected end in tag"); return nullptr; }
            if (s[i] == '/') {
                if (i + 1 < s.size() && s[i + 1] == '>') { i += 2; return e; }
                Fail("stray '/'"); return nullptr;
            }
            if (s[i] == '>') { i++; break; }
            std::string key;
            while (i < s.size() && NameChar(s[i])) key += s[i++];
            if (key.empty()) { Fail("bad attribute name"); return nullptr; }
            SkipSpace();
            if (i >= s.size() || s[i] != '=') { Fail("expected '='"); return nullptr; }
            i++;
            SkipSpace();
            if (i >= s.size() || (s[i] != '"' && s[i] != '\'')) { Fail("expected quoted value"); return nullptr; }
            const char q = s[i++];
            size_t e2 = s.find(q, i);
            if (e2 == std::string::npos) { Fail("unterminated attribute value"); return nullptr; }
            e->attrs.emplace_back(key, s.substr(i, e2 - i));
            i = e2 + 1;
        }
        // content: child elements, comments, text (ignored) up to the closing tag
        for (;;) {
            while (i < s.size() && s[i] != '<') i++;
            if (i >= s.size()) { Fail("missing closing tag"); return nullptr; }
            if (!SkipMisc()) return nullptr;
            if (i >= s.size()) { Fail("missing closing tag"); return nullptr; }
            if (s[i] != '<') continue;
            if (i + 1 < s.size() && s[i + 1] == '/') {
                size_t e2 = s.find('>', i);
                if (e2 == std::string::npos) { Fail("unterminated closing tag"); return nullptr; }
                std::string close = s.substr(i + 2, e2 - i - 2);
                while (!close.empty() && isspace((unsigned char)close.back())) close.pop_back();
                if (close != e->name) { Fail("mismatched closing tag"); return nullptr; }
                i = e2 + 1;
                return e;
            }
            std::unique_ptr<XmlElement> c = ParseElement();
            if (!c) return nullptr;
            e->children.push_back(std::move(c));
        }
    }
};

bool Is(const XmlElement &e, const char *name) { return strcasecmp(e.name.c_str(), name) == 0; }
bool IsStr(const char *a, const char *b) { return a && strcasecmp(a, b) == 0; }

// ReadFloat / ReadVector / ReadColor, FIN/xmlload.cpp:453-496
void ReadFloat(const XmlElement &e, float &f, const char *name = "value")
{
    double d = (double)f;
    e.QueryDouble(name, &d);
    f = (float)d;
}
void ReadVector(const XmlElement &e, Point3 &v)
{
    double x = v.x, y = v.y, z = v.z;
    e.QueryDouble("x", &x); e.QueryDouble("y", &y); e.QueryDouble("z", &z);
    v = Point3((float)x, (float)y, (float)z);
    float f = 1;
    ReadFloat(e, f);
    v = v * f;
}
void ReadColor(const XmlElement &e, Color &c)
{
    double r = c.r, g = c.g, b = c.b;
    e.QueryDouble("r", &r); e.QueryDouble("g", &g); e.QueryDouble("b", &b);
    c = Color((float)r, (float)g, (float)b);
    float f = 1;
    ReadFloat(e, f);
    c *= f;
}

struct Loader {
    Scene &sc;
    std::string dir;             // directory of the XML file: OBJ names resolve against it
    std::string *err;
    bool ok = true;
    std::vector<std::pair<Node *, std::string>> nodeMtl;

    void Error(const std::string &m) { ok = false; if (err && err->empty()) *err = m; }

    // ReadTexture(const char*), FIN/xmlload.cpp:535-554: an image shared through the TextureList
    Texture *ReadTextureFile(const char *texName)
    {
        Texture *tex = sc.FindTexture(texName);
        if (!tex) {
            std::unique_ptr<TextureFile> f(new TextureFile);
            const std::string n = texName;
            f->name = (!n.empty() && n[0] == '/') ? n : dir + n;
            std::string e2;
            if (!f->Load(&e2)) fprintf(stderr, "rt_mi355x: cannot load texture \"%s\": %s\n", f->name.c_str(), e2.c_str());   // map stays, samples black
            else { f->name = texName; tex = f.get(); sc.textureList.push_back(std::move(f)); }
        }
        return tex;
    }

    // ReadTexture(TiXmlElement*), FIN/xmlload.cpp:500-530: attribute texture="checkerboard" (children
    // color1/color2) or a PNG/PPM file name shared through the TextureList; the element's own
    // scale/rotate/translate children transform the map
    TextureMap *ReadTexture(const XmlElement &e)
    {
        const char *texName = e.Attribute("texture");
        if (!texName) return nullptr;
        Texture *tex = nullptr;
        if (IsStr(texName, "checkerboard")) {
            std::unique_ptr<TextureChecker> c(new TextureChecker);
            for (auto &ch : e.children) {
                if (Is(*ch, "color1")) { Color col(0, 0, 0); ReadColor(*ch, col); c->SetColor1(col); }
                else if (Is(*ch, "color2")) { Color col(0, 0, 0); ReadColor(*ch, col); c->SetColor2(col); }
            }
            c->name = texName;
            tex = c.get();
            sc.textureList.push_back(std::move(c));
        } else tex = ReadTextureFile(texName);
        TextureMap *map = new TextureMap(tex);
        LoadTransform(*map, e);
        return map;
    }

    // LoadTransform, FIN/xmlload.cpp:265-291
    void LoadTransform(Transformation &t, const XmlElement &e)
    {
        for (auto &c : e.children) {
            if (Is(*c, "scale")) { Point3 s(1, 1, 1); ReadVector(*c, s); t.Scale(s.x, s.y, s.z); }
            else if (Is(*c, "rotate")) {
                Point3 s(0, 0, 0);
                ReadVector(*c, s);
                s.Normalize();
                float a = 0;
                ReadFloat(*c, a, "angle");
                t.Rotate(s, a);
            } else if (Is(*c, "translate")) { Point3 p(0, 0, 0); ReadVector(*c, p); t.Translate(p); }
        }
    }

    // LoadNode, FIN/xmlload.cpp:168-261
    void LoadNode(Node &parent, const XmlElement &e)
    {
        Node *node = new Node;
        parent.AppendChild(node);
        const char *name = e.Attribute("name");
        if (name) node->name = name;
        const char *mtlName = e.Attribute("material");
        if (mtlName) { node->mtlName = mtlName; nodeMtl.emplace_back(node, mtlName); }
        const char *type = e.Attribute("type");
        if (type) {
            if (IsStr(type, "sphere")) node->SetNodeObj(&sc.theSphere);
            else if (IsStr(type, "plane")) node->SetNodeObj(&sc.thePlane);
            else if (IsStr(type, "obj")) {
                const std::string key = name ? name : "";
                TriObj *obj = sc.FindObj(key);           // meshes are shared by file name (:202-203)
                if (!obj) {
                    std::unique_ptr<TriObj> t(new TriObj);
                    std::string e2;
                    const std::string path = (!key.empty() && key[0] == '/') ? key : dir + key;
                    if (!t->Load(path.c_str(), mtlName == nullptr, &e2)) {
                        // the reference prints an error and leaves the node without an object (:205-207)
                        fprintf(stderr, "rt_mi355x: cannot load OBJ \"%s\": %s\n", path.c_str(), e2.c_str());
                    } else {
                        obj = t.get();
                        sc.objList.emplace_back(key, std::move(t));
                        if (obj->NM() > 0 && !sc.FindMaterial(key)) MakeMultiMtl(*obj, key, node);
                    }
                }
                node->SetNodeObj(obj);
            }
            // unknown type: node without an object, as in the reference (:244-246)
        }
        for (auto &c : e.children) if (Is(*c, "object")) LoadNode(*node, *c);
        LoadTransform(*node, e);
    }

    // the multi-material LoadNode generates for an OBJ that brought its own .mtl
    // (FIN/xmlload.cpp:211-240), including that a map_Ks REPLACES the diffuse texture (:223-224;
    // the reflection texture of :227 is never sampled by Shade).  Texture names resolve like the
    // XML's own: relative to the scene file's directory.
    void MakeMultiMtl(const TriObj &obj, const std::string &name, Node *node)
    {
        std::unique_ptr<MultiMtl> mm(new MultiMtl);
        for (unsigned i = 0; i < obj.NM(); i++) {
            const ObjMtl &mtl = obj.mtls[i];
            MtlBlinn *m = new MtlBlinn;
            m->SetDiffuse(Color(mtl.Kd[0], mtl.Kd[1], mtl.Kd[2]));
            m->SetSpecular(Color(mtl.Ks[0], mtl.Ks[1], mtl.Ks[2]));
            m->SetGlossiness(mtl.Ns);
            m->SetRefractionIndex(mtl.Ni);
            if (!mtl.map_Kd.empty()) m->SetDiffuseTexture(new TextureMap(ReadTextureFile(mtl.map_Kd.c_str())));
            if (!mtl.map_Ks.empty()) m->SetDiffuseTexture(new TextureMap(ReadTextureFile(mtl.map_Ks.c_str())));
            if (mtl.illum > 2 && mtl.illum <= 7) {
                m->SetReflection(Color(mtl.Ks[0], mtl.Ks[1], mtl.Ks[2]));
                if (mtl.illum >= 6) m->SetRefraction(Color(1 - mtl.Tf[0], 1 - mtl.Tf[1], 1 - mtl.Tf[2]));
            }
            mm->AppendMaterial(m);
        }
        mm->name = name;
        sc.materials.push_back(std::move(mm));
        node->mtlName = name;
        nodeMtl.emplace_back(node, name);
    }

    // LoadMaterial, FIN/xmlload.cpp:295-371 (textures are not read on this path)
    void LoadMaterial(const XmlElement &e)
    {
        const char *type = e.Attribute("type");
        if (!IsStr(type, "blinn")) return;
        std::unique_ptr<MtlBlinn> m(new MtlBlinn);
        for (auto &c : e.children) {
            Color col(1, 1, 1);
            float f = 1;
            if (Is(*c, "diffuse")) { ReadColor(*c, col); m->SetDiffuse(col); m->SetDiffuseTexture(ReadTexture(*c)); }
            else if (Is(*c, "specular")) { ReadColor(*c, col); m->SetSpecular(col); m->SetSpecularTexture(ReadTexture(*c)); }
            else if (Is(*c, "glossiness")) { ReadFloat(*c, f); m->SetGlossiness(f); }
            else if (Is(*c, "emission")) { ReadColor(*c, col); m->SetEmission(col); }
            else if (Is(*c, "reflection")) {
                ReadColor(*c, col); m->SetReflection(col);
                f = 0; ReadFloat(*c, f, "glossiness"); m->SetReflectionGlossiness(f);
            } else if (Is(*c, "refraction")) {
                ReadColor(*c, col); m->SetRefraction(col);
                ReadFloat(*c, f, "index"); m->SetRefractionIndex(f);
                f = 0; ReadFloat(*c, f, "glossiness"); m->SetRefractionGlossiness(f);
            } else if (Is(*c, "absorption")) { ReadColor(*c, col); m->SetAbsorption(col); }
        }
        const char *name = e.Attribute("name");
        if (name) m->name = name;
        sc.materials.push_back(std::move(m));
    }

    // LoadLight, FIN/xmlload.cpp:375-449
    void LoadLight(const XmlElement &e)
    {
        const char *type = e.Attribute("type");
        std::unique_ptr<Light> light;
        if (IsStr(type, "ambient")) {
            std::unique_ptr<AmbientLight> l(new AmbientLight);
            for (auto &c : e.children) if (Is(*c, "intensity")) { Color col(1, 1, 1); ReadColor(*c, col); l->SetIntensity(col); }
            light = std::move(l);
        } else if (IsStr(type, "direct")) {
            std::unique_ptr<DirectLight> l(new DirectLight);
            for (auto &c : e.children) {
                if (Is(*c, "intensity")) { Color col(1, 1, 1); ReadColor(*c, col); l->SetIntensity(col); }
                else if (Is(*c, "direction")) { Point3 v(1, 1, 1); ReadVector(*c, v); l->SetDirection(v); }
            }
            light = std::move(l);
        } else if (IsStr(type, "point")) {
            std::unique_ptr<PointLight> l(new PointLight);
            for (auto &c : e.children) {
                if (Is(*c, "intensity")) { Color col(1, 1, 1); ReadColor(*c, col); l->SetIntensity(col); }
                else if (Is(*c, "position")) { Point3 v(0, 0, 0); ReadVector(*c, v); l->SetPosition(v); }
                else if (Is(*c, "size")) { float f = 0; ReadFloat(*c, f); l->SetSize(f); }
            }
            light = std::move(l);
        }
        if (light) {
            const char *name = e.Attribute("name");
            if (name) light->name = name;
            sc.lights.push_back(std::move(light));
        }
    }

    // LoadScene(TiXmlElement*), FIN/xmlload.cpp:140-164
    void LoadSceneElement(const XmlElement &e)
    {
        for (auto &c : e.children) {
            if (Is(*c, "background")) { Color col(1, 1, 1); ReadColor(*c, col); sc.background.SetColor(col); sc.background.SetTexture(ReadTexture(*c)); }
            else if (Is(*c, "environment")) { Color col(1, 1, 1); ReadColor(*c, col); sc.environment.SetColor(col); sc.environment.SetTexture(ReadTexture(*c)); }
            else if (Is(*c, "object")) LoadNode(sc.rootNode, *c);
            else if (Is(*c, "material")) LoadMaterial(*c);
            else if (Is(*c, "light")) LoadLight(*c);
        }
    }
};

}  // namespace

int LoadScene(Scene &scene, const char *filename, std::string *err)
{
    std::string local;
    if (!err) err = &local;
    err->clear();
    FILE *fp = fopen(filename, "rb");
    if (!fp) { *err = std::string("Failed to load the file \"") + filename + "\""; return 0; }
    std::string text;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) text.append(buf, n);
    fclose(fp);
    XmlParser parser(text, err);
    std::unique_ptr<XmlElement> xml = parser.ParseElement();
    if (!xml) return 0;
    if (xml->name != "xml") { *err = "No \"xml\" tag found."; return 0; }
    const XmlElement *sceneEl = xml->FirstChild("scene");
    if (!sceneEl) { *err = "No \"scene\" tag found."; return 0; }
    const XmlElement *cam = xml->FirstChild("camera");
    if (!cam) { *err = "No \"camera\" tag found."; return 0; }

    scene.Clear();
    Loader L{scene, "", err};
    std::string fn(filename);
    size_t slash = fn.find_last_of('/');
    if (slash != std::string::npos) L.dir = fn.substr(0, slash + 1);
    L.LoadSceneElement(*sceneEl);
    if (!L.ok) return 0;
    for (auto &nm : L.nodeMtl) if (Material *m = scene.FindMaterial(nm.second)) nm.first->SetMaterial(m);

    // camera, FIN/xmlload.cpp:107-127
    Camera &camera = scene.camera;
    camera.Init();
    camera.dir += camera.pos;
    for (auto &c : cam->children) {
        if (Is(*c, "position")) ReadVector(*c, camera.pos);
        else if (Is(*c, "target")) ReadVector(*c, camera.dir);
        else if (Is(*c, "up")) ReadVector(*c, camera.up);
        else if (Is(*c, "fov")) ReadFloat(*c, camera.fov);
        else if (Is(*c, "focaldist")) ReadFloat(*c, camera.focaldist);
        else if (Is(*c, "dof")) ReadFloat(*c, camera.dof);
        else if (Is(*c, "width")) c->QueryInt("value", &camera.imgWidth);
        else if (Is(*c, "height")) c->QueryInt("value", &camera.imgHeight);
    }
    camera.dir = camera.dir - camera.pos;
    camera.dir.Normalize();
    Point3 x = camera.dir ^ camera.up;
    camera.up = (x ^ camera.dir).GetNormalized();
    return 1;
}

}  // namespace rt
