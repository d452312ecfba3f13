/*
 * TEST INFRASTRUCTURE (build container only; nothing in the product, the GPU tests or bench.py uses this).
 *
 * A headless OpenGL 4.5 core context on Mesa's software rasteriser (llvmpipe, swrast_dri.so of the image's Mesa 23.2.1) without an X
 * server, EGL or OSMesa -- none of which the image has -- by speaking the driver's DRI "swrast" loader interface directly
 * (/usr/include/GL/internal/dri_interface.h), and a small C API over it for tests/golden/make_mesa_vectors.py: compile a fragment shader,
 * set uniforms and textures, draw one full-screen triangle into an RGBA32F target, read the floats back.
 *
 * Why: Mesa's GLSL compiler and llvmpipe are a GLSL implementation the author of this repository did not write.  The reference's shader
 * text, run through it, is a second pin of the oracle beside the repository's own interpreter (tests/golden/gdshader_vm.py).
 *
 *   gcc -O2 -shared -fPIC -o libmesa_glsl_runner.so mesa_glsl_runner.c -ldl
 */
#include <GL/gl.h>
#include <GL/glext.h>
#include <GL/internal/dri_interface.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const __DRIcoreExtension *g_core;
static const __DRIswrastExtension *g_swrast;
static __DRIscreen *g_screen;
static __DRIcontext *g_ctx;
static __DRIdrawable *g_draw;
static void *(*g_get_proc)(const char *);
static char g_err[512];

/* the window-system drawable the DRI interface insists on: a 16 x 16 nothing (everything is drawn into a framebuffer object) */
static void cb_get_drawable_info(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *priv) { *x = *y = 0; *w = *h = 16; }
static void cb_put_image(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *priv) {}
static void cb_get_image(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *priv) { memset(data, 0, (size_t)w * h * 4); }
static void cb_put_image2(__DRIdrawable *d, int op, int x, int y, int w, int h, int stride, char *data, void *priv) {}
static void cb_get_image2(__DRIdrawable *d, int x, int y, int w, int h, int stride, char *data, void *priv) {
    for (int r = 0; r < h; ++r) memset(data + (size_t)r * stride, 0, (size_t)w * 4);
}
static const __DRIswrastLoaderExtension g_loader = {
    .base = {__DRI_SWRAST_LOADER, 3},
    .getDrawableInfo = cb_get_drawable_info,
    .putImage = cb_put_image,
    .getImage = cb_get_image,
    .putImage2 = cb_put_image2,
    .getImage2 = cb_get_image2,
};
static const __DRIextension *g_loader_exts[] = {&g_loader.base, NULL};

#define GLF(type, name) static type p_##name;
#define GL_FUNCS(X)                                                                                                                      \
    X(PFNGLCREATESHADERPROC, glCreateShader) X(PFNGLSHADERSOURCEPROC, glShaderSource) X(PFNGLCOMPILESHADERPROC, glCompileShader)           \
    X(PFNGLGETSHADERIVPROC, glGetShaderiv) X(PFNGLGETSHADERINFOLOGPROC, glGetShaderInfoLog) X(PFNGLCREATEPROGRAMPROC, glCreateProgram)     \
    X(PFNGLATTACHSHADERPROC, glAttachShader) X(PFNGLLINKPROGRAMPROC, glLinkProgram) X(PFNGLGETPROGRAMIVPROC, glGetProgramiv)               \
    X(PFNGLGETPROGRAMINFOLOGPROC, glGetProgramInfoLog) X(PFNGLUSEPROGRAMPROC, glUseProgram) X(PFNGLDELETEPROGRAMPROC, glDeleteProgram)     \
    X(PFNGLDELETESHADERPROC, glDeleteShader) X(PFNGLGETUNIFORMLOCATIONPROC, glGetUniformLocation) X(PFNGLUNIFORM1FVPROC, glUniform1fv)     \
    X(PFNGLUNIFORM2FVPROC, glUniform2fv) X(PFNGLUNIFORM3FVPROC, glUniform3fv) X(PFNGLUNIFORM4FVPROC, glUniform4fv)                         \
    X(PFNGLUNIFORM1IPROC, glUniform1i) X(PFNGLUNIFORMMATRIX2FVPROC, glUniformMatrix2fv) X(PFNGLUNIFORMMATRIX3FVPROC, glUniformMatrix3fv)   \
    X(PFNGLUNIFORMMATRIX4FVPROC, glUniformMatrix4fv) X(PFNGLGENFRAMEBUFFERSPROC, glGenFramebuffers)                                        \
    X(PFNGLBINDFRAMEBUFFERPROC, glBindFramebuffer) X(PFNGLFRAMEBUFFERTEXTURE2DPROC, glFramebufferTexture2D)                                \
    X(PFNGLCHECKFRAMEBUFFERSTATUSPROC, glCheckFramebufferStatus) X(PFNGLDELETEFRAMEBUFFERSPROC, glDeleteFramebuffers)                      \
    X(PFNGLGENVERTEXARRAYSPROC, glGenVertexArrays) X(PFNGLBINDVERTEXARRAYPROC, glBindVertexArray) X(PFNGLACTIVETEXTUREPROC, glActiveTexture) \
    X(PFNGLTEXIMAGE3DPROC, glTexImage3D) X(PFNGLGENERATEMIPMAPPROC, glGenerateMipmap) X(PFNGLDRAWBUFFERSPROC, glDrawBuffers) \
    X(PFNGLBLENDFUNCSEPARATEPROC, glBlendFuncSeparate) X(PFNGLBLENDEQUATIONPROC, glBlendEquation)
GL_FUNCS(GLF)
/* GL 1.x entry points come through the same table: libGL.so here is libglvnd's, whose dispatch this context is not registered with */
static void (*p_glGenTextures)(GLsizei, GLuint *);
static void (*p_glBindTexture)(GLenum, GLuint);
static void (*p_glTexImage2D)(GLenum, GLint, GLint, GLsizei, GLsizei, GLint, GLenum, GLenum, const void *);
static void (*p_glTexParameteri)(GLenum, GLenum, GLint);
static void (*p_glDeleteTextures)(GLsizei, const GLuint *);
static void (*p_glViewport)(GLint, GLint, GLsizei, GLsizei);
static void (*p_glDrawArrays)(GLenum, GLint, GLsizei);
static void (*p_glReadPixels)(GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, void *);
static void (*p_glFinish)(void);
static GLenum (*p_glGetError)(void);
static const GLubyte *(*p_glGetString)(GLenum);
static void (*p_glPixelStorei)(GLenum, GLint);
static void (*p_glDisable)(GLenum);
static void (*p_glClearColor)(GLfloat, GLfloat, GLfloat, GLfloat);
static void (*p_glClear)(GLbitfield);
static void (*p_glEnable)(GLenum);

const char *mgl_error(void) { return g_err; }

static int fail(const char *msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return -1;
}

int mgl_init(const char *driver_path) {
    if (g_ctx) return 0;
    void *glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!glapi) return fail("libglapi.so.0 not found");
    g_get_proc = (void *(*)(const char *))dlsym(glapi, "_glapi_get_proc_address");
    if (!g_get_proc) return fail("_glapi_get_proc_address not found");
    void *drv = dlopen(driver_path ? driver_path : "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", RTLD_NOW | RTLD_GLOBAL);
    if (!drv) return fail(dlerror());
    const __DRIextension **(*get_exts)(void) = (const __DRIextension **(*)(void))dlsym(drv, "__driDriverGetExtensions_swrast");
    if (!get_exts) return fail("__driDriverGetExtensions_swrast not found");
    const __DRIextension **exts = get_exts();
    for (int i = 0; exts[i]; ++i) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) g_core = (const __DRIcoreExtension *)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) g_swrast = (const __DRIswrastExtension *)exts[i];
    }
    if (!g_core || !g_swrast || g_swrast->base.version < 4) return fail("driver lacks DRI_Core / DRI_SWRast v4");
    const __DRIconfig **configs = NULL;
    g_screen = g_swrast->createNewScreen2(0, g_loader_exts, exts, &configs, NULL);
    if (!g_screen || !configs || !configs[0]) return fail("createNewScreen2 failed");
    const __DRIconfig *cfg = configs[0];
    for (int i = 0; configs[i]; ++i) {  /* an RGBA8 single-sample config; anything will do, the drawable is never looked at */
        unsigned r = 0, db = 0;
        g_core->getConfigAttrib(configs[i], __DRI_ATTRIB_RED_SIZE, &r);
        g_core->getConfigAttrib(configs[i], __DRI_ATTRIB_DOUBLE_BUFFER, &db);
        if (r == 8 && !db) { cfg = configs[i]; break; }
    }
    unsigned err = 0;
    const uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 5};
    g_ctx = g_swrast->createContextAttribs(g_screen, __DRI_API_OPENGL_CORE, cfg, NULL, 2, attribs, &err, NULL);
    if (!g_ctx) { snprintf(g_err, sizeof g_err, "createContextAttribs failed (%u)", err); return -1; }
    g_draw = g_swrast->createNewDrawable(g_screen, cfg, NULL);
    if (!g_draw) return fail("createNewDrawable failed");
    if (!g_core->bindContext(g_ctx, g_draw, g_draw)) return fail("bindContext failed");
#define LOAD(type, name) if (!(p_##name = (type)g_get_proc(#name))) return fail("missing " #name);
    GL_FUNCS(LOAD)
#define LOAD1(name) if (!(*(void **)&p_##name = g_get_proc(#name))) return fail("missing " #name);
    LOAD1(glGenTextures) LOAD1(glBindTexture) LOAD1(glTexImage2D) LOAD1(glTexParameteri) LOAD1(glDeleteTextures) LOAD1(glViewport)
    LOAD1(glDrawArrays) LOAD1(glReadPixels) LOAD1(glFinish) LOAD1(glGetError) LOAD1(glGetString) LOAD1(glPixelStorei) LOAD1(glDisable)
    LOAD1(glClearColor) LOAD1(glClear) LOAD1(glEnable)
    GLuint vao;
    p_glGenVertexArrays(1, &vao);
    p_glBindVertexArray(vao);
    p_glPixelStorei(GL_UNPACK_ALIGNMENT, 1);
    p_glPixelStorei(GL_PACK_ALIGNMENT, 1);
    p_glDisable(GL_BLEND);
    p_glDisable(GL_DEPTH_TEST);
    p_glDisable(GL_DITHER);
    p_glEnable(GL_TEXTURE_CUBE_MAP_SEAMLESS);
    return 0;
}

/* "vendor | renderer | version" */
const char *mgl_info(void) {
    static char s[512];
    snprintf(s, sizeof s, "%s | %s | %s | GLSL %s", p_glGetString(GL_VENDOR), p_glGetString(GL_RENDERER), p_glGetString(GL_VERSION),
             p_glGetString(GL_SHADING_LANGUAGE_VERSION));
    return s;
}

static GLuint compile(GLenum kind, const char *src, char *log, int logsz) {
    GLuint sh = p_glCreateShader(kind);
    p_glShaderSource(sh, 1, &src, NULL);
    p_glCompileShader(sh);
    GLint ok = 0;
    p_glGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
    if (!ok) {
        p_glGetShaderInfoLog(sh, logsz, NULL, log);
        p_glDeleteShader(sh);
        return 0;
    }
    return sh;
}

static const char *VS =
    "#version 450 core\n"
    "void main() { vec2 p = vec2((gl_VertexID & 1) * 4 - 1, (gl_VertexID >> 1) * 4 - 1); gl_Position = vec4(p, 0.0, 1.0); }\n";

/* fragment shader -> program (0 and the compiler's log on failure) */
unsigned mgl_program(const char *fs_src, char *log, int logsz) {
    if (logsz > 0) log[0] = 0;
    GLuint vs = compile(GL_VERTEX_SHADER, VS, log, logsz);
    if (!vs) return 0;
    GLuint fs = compile(GL_FRAGMENT_SHADER, fs_src, log, logsz);
    if (!fs) {
        p_glDeleteShader(vs);
        return 0;
    }
    GLuint prog = p_glCreateProgram();
    p_glAttachShader(prog, vs);
    p_glAttachShader(prog, fs);
    p_glLinkProgram(prog);
    p_glDeleteShader(vs);   /* flagged for deletion; freed with the program */
    p_glDeleteShader(fs);
    GLint ok = 0;
    p_glGetProgramiv(prog, GL_LINK_STATUS, &ok);
    if (!ok) {
        p_glGetProgramInfoLog(prog, logsz, NULL, log);
        p_glDeleteProgram(prog);
        return 0;
    }
    p_glUseProgram(prog);
    return prog;
}

/* kind: 1..4 = float / vec2 / vec3 / vec4, 9 = mat3 (column-major), 16 = mat4 (column-major), 22 = mat2, 0 = int (value in v[0]); -1 if not active */
int mgl_uniform(unsigned prog, const char *name, int kind, const float *v) {
    p_glUseProgram(prog);
    GLint loc = p_glGetUniformLocation(prog, name);
    if (loc < 0) return -1;
    switch (kind) {
    case 0: p_glUniform1i(loc, (int)v[0]); break;
    case 1: p_glUniform1fv(loc, 1, v); break;
    case 2: p_glUniform2fv(loc, 1, v); break;
    case 3: p_glUniform3fv(loc, 1, v); break;
    case 4: p_glUniform4fv(loc, 1, v); break;
    case 9: p_glUniformMatrix3fv(loc, 1, GL_FALSE, v); break;
    case 16: p_glUniformMatrix4fv(loc, 1, GL_FALSE, v); break;
    case 22: p_glUniformMatrix2fv(loc, 1, GL_FALSE, v); break;
    default: return -2;
    }
    return p_glGetError() == GL_NO_ERROR ? 0 : -3;
}

/* dims: 2 = 2D (w x h), 3 = 3D (w x h x d), 6 = cube (6 faces of w x w, face order +X -X +Y -Y +Z -Z, d = number of mip levels supplied
 * back to back, each level's six faces together).  fmt: 0 = R8 unorm, 1 = RGBA8 unorm, 2 = R32F.  filter: 0 nearest, 1 linear,
 * 2 linear-mipmap-linear.  repeat: 0 clamp-to-edge, 1 repeat.  Binds the texture to `unit` and sets the sampler uniform. */
unsigned mgl_texture(unsigned prog, const char *sampler, int unit, int dims, int w, int h, int d, int fmt, int filter, int repeat,
                     const void *data) {
    static const GLenum ifmt[] = {GL_R8, GL_RGBA8, GL_R32F}, efmt[] = {GL_RED, GL_RGBA, GL_RED}, etype[] = {GL_UNSIGNED_BYTE, GL_UNSIGNED_BYTE, GL_FLOAT};
    static const int bpp[] = {1, 4, 4};
    const GLenum target = dims == 2 ? GL_TEXTURE_2D : dims == 3 ? GL_TEXTURE_3D : GL_TEXTURE_CUBE_MAP;
    GLuint tex;
    p_glGenTextures(1, &tex);
    p_glActiveTexture(GL_TEXTURE0 + unit);
    p_glBindTexture(target, tex);
    int levels = 1;
    if (dims == 2) {
        p_glTexImage2D(GL_TEXTURE_2D, 0, ifmt[fmt], w, h, 0, efmt[fmt], etype[fmt], data);
    } else if (dims == 3) {
        p_glTexImage3D(GL_TEXTURE_3D, 0, ifmt[fmt], w, h, d, 0, efmt[fmt], etype[fmt], data);
    } else {
        const unsigned char *p = (const unsigned char *)data;
        levels = d;
        for (int l = 0, n = w; l < levels; ++l, n = n > 1 ? n / 2 : 1) {
            for (int f = 0; f < 6; ++f) {
                p_glTexImage2D(GL_TEXTURE_CUBE_MAP_POSITIVE_X + f, l, ifmt[fmt], n, n, 0, efmt[fmt], etype[fmt], p);
                p += (size_t)n * n * bpp[fmt];
            }
        }
        p_glTexParameteri(target, GL_TEXTURE_MAX_LEVEL, levels - 1);
    }
    const GLenum minf = filter == 0 ? GL_NEAREST : filter == 1 ? GL_LINEAR : GL_LINEAR_MIPMAP_LINEAR;
    p_glTexParameteri(target, GL_TEXTURE_MIN_FILTER, minf);
    p_glTexParameteri(target, GL_TEXTURE_MAG_FILTER, filter == 0 ? GL_NEAREST : GL_LINEAR);
    const GLenum wrap = repeat ? GL_REPEAT : GL_CLAMP_TO_EDGE;
    p_glTexParameteri(target, GL_TEXTURE_WRAP_S, wrap);
    p_glTexParameteri(target, GL_TEXTURE_WRAP_T, wrap);
    if (dims != 2) p_glTexParameteri(target, GL_TEXTURE_WRAP_R, wrap);
    p_glUseProgram(prog);
    GLint loc = p_glGetUniformLocation(prog, sampler);
    if (loc >= 0) p_glUniform1i(loc, unit);
    if (p_glGetError() != GL_NO_ERROR) return 0;
    return tex;
}

void mgl_delete_texture(unsigned tex) { p_glDeleteTextures(1, &tex); }
void mgl_delete_program(unsigned prog) { p_glDeleteProgram(prog); }

/* one full-screen triangle into a w x h RGBA32F target; rgba: w * h * 4 floats, row 0 = the BOTTOM row (gl_FragCoord.y = 0.5).
 * dst == NULL: the target is cleared to `clear`, no blending.  dst != NULL: the target starts as dst (the scene's colour buffer) and the fragment's
 * (rgb, a) goes through the FIXED-FUNCTION blend stage as a blend_mix material's does: colour SRC_ALPHA / ONE_MINUS_SRC_ALPHA, alpha ONE /
 * ONE_MINUS_SRC_ALPHA, equation ADD; a discarded fragment leaves the target as it was. */
int mgl_draw_over(unsigned prog, int w, int h, float clear, const float *dst, float *rgba) {
    GLuint fbo, tex;
    p_glGenTextures(1, &tex);
    p_glActiveTexture(GL_TEXTURE0 + 15);
    p_glBindTexture(GL_TEXTURE_2D, tex);
    p_glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, w, h, 0, GL_RGBA, GL_FLOAT, dst);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    p_glGenFramebuffers(1, &fbo);
    p_glBindFramebuffer(GL_FRAMEBUFFER, fbo);
    p_glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, tex, 0);
    if (p_glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) {
        p_glBindFramebuffer(GL_FRAMEBUFFER, 0);
        p_glDeleteFramebuffers(1, &fbo);
        p_glDeleteTextures(1, &tex);
        return fail("framebuffer incomplete");
    }
    const GLenum bufs[] = {GL_COLOR_ATTACHMENT0};
    p_glDrawBuffers(1, bufs);
    p_glViewport(0, 0, w, h);
    if (dst) {
        p_glEnable(GL_BLEND);
        p_glBlendEquation(GL_FUNC_ADD);
        p_glBlendFuncSeparate(GL_SRC_ALPHA, GL_ONE_MINUS_SRC_ALPHA, GL_ONE, GL_ONE_MINUS_SRC_ALPHA);
    } else {
        p_glClearColor(clear, clear, clear, clear);
        p_glClear(GL_COLOR_BUFFER_BIT);
    }
    p_glUseProgram(prog);
    p_glDrawArrays(GL_TRIANGLES, 0, 3);
    p_glFinish();
    p_glDisable(GL_BLEND);
    p_glReadPixels(0, 0, w, h, GL_RGBA, GL_FLOAT, rgba);
    const GLenum e = p_glGetError();
    p_glBindFramebuffer(GL_FRAMEBUFFER, 0);
    p_glDeleteFramebuffers(1, &fbo);
    p_glDeleteTextures(1, &tex);
    if (e != GL_NO_ERROR) { snprintf(g_err, sizeof g_err, "GL error 0x%x", e); return -1; }
    return 0;
}

int mgl_draw(unsigned prog, int w, int h, float clear, float *rgba) { return mgl_draw_over(prog, w, h, clear, NULL, rgba); }
