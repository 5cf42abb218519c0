/* gl_ref_mesa.c -- test infrastructure, build container only: the two passes of Utils/Render_utils.py drawn by a DESKTOP
 * OpenGL implementation (Mesa 23.2 llvmpipe, OpenGL 4.5 core, GLSL 4.50, aliased line widths 1..255), so that the
 * reference's default 3-pixel lines (Render_utils.py:28) and its own desktop-GLSL shader text can be used -- what
 * SwiftShader (ES 3.0, width range [1,1], tools/gl_ref/gl_ref.c) cannot do.
 *
 * The image has no X server, no Mesa EGL and no OSMesa; what it has is Mesa's software DRI driver (swrast_dri.so) and the
 * DRI interface header of mesa-common-dev.  This program is its own minimal DRI *loader*: it asks the driver for its
 * DRI_Core / DRI_SWRast extensions (__driDriverGetExtensions_swrast), creates a screen with a DRI_SWRastLoader extension
 * whose put/get-image callbacks do nothing (everything is drawn into a framebuffer object), a 3.3 core context, a dummy
 * drawable, binds them and resolves GL entry points through libglapi (_glapi_get_proc_address).
 *
 * The shader sources are NOT in this file: the generator (tools/gen_golden_gl_mesa.py) cuts the four GLSL strings out of
 * /root/reference/Utils/Render_utils.py at generation time and hands them over in a text file; they are compiled as they
 * are.  Uniform / attribute names are therefore the reference's: projection, transform, depthOption (triangles),
 * colorOption (lines); vertexPosition / LinePosition at location 0, Tangent at location 1.
 * State as the reference sets it (Renderer.__init__ :203-240): RGB32F colour renderbuffer + 24-bit depth attachment,
 * viewport = buffer size, clear colour from the job, depth test LESS, no culling, line width from the job.
 *
 *   gl_ref_mesa <job.bin> <shaders.txt> <out.bin>
 * job.bin: as for gl_ref.c.  shaders.txt: triangle VS, triangle FS, line VS, line FS, separated by lines "=====".
 * out.bin: float32 line_width_range[2]; float32 RGBA [H][W], rows bottom-up as glReadPixels returns them.            */
#define _GNU_SOURCE
#include <GL/gl.h>
#include <GL/glext.h>
#include <GL/internal/dri_interface.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void get_info(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p) { *x = *y = 0; *w = *h = 16; }
static void put_image(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p) {}
static void get_image(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p) { memset(data, 0, (size_t)w * h * 4); }
static void put_image2(__DRIdrawable *d, int op, int x, int y, int w, int h, int stride, char *data, void *p) {}
static void get_image2(__DRIdrawable *d, int x, int y, int w, int h, int stride, char *data, void *p) { memset(data, 0, (size_t)stride * h); }
static const __DRIswrastLoaderExtension loader = {.base = {__DRI_SWRAST_LOADER, 3}, .getDrawableInfo = get_info,
                                                  .putImage = put_image, .getImage = get_image, .putImage2 = put_image2,
                                                  .getImage2 = get_image2};
static const __DRIextension *loader_ext[] = {&loader.base, NULL};

static void *(*gpa)(const char *);
#define GLF(type, name) type name = (type)gpa(#name); if (!name) { fprintf(stderr, "no %s\n", #name); return 2; }

static void rd(void *dst, size_t n, FILE *f) {
    if (fread(dst, 1, n, f) != n) { fprintf(stderr, "short job file\n"); exit(4); }
}

int main(int argc, char **argv) {
    if (argc < 4) return 1;
    void *h = dlopen("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    const __DRIextension **(*get)(void) = dlsym(h, "__driDriverGetExtensions_swrast");
    if (!get) return 2;
    const __DRIextension **exts = get();
    const __DRIcoreExtension *core = NULL;
    const __DRIswrastExtension *sw = NULL;
    for (int i = 0; exts[i]; ++i) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) core = (const void *)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) sw = (const void *)exts[i];
    }
    if (!core || !sw || sw->base.version < 4) return 2;
    const __DRIconfig **configs = NULL;
    __DRIscreen *scr = sw->createNewScreen2(0, loader_ext, exts, &configs, NULL);
    if (!scr) return 3;
    const __DRIconfig *cfg = NULL;
    for (int i = 0; configs[i]; ++i) {
        unsigned r = 0, a = 0, d = 0, db = 1;
        core->getConfigAttrib(configs[i], __DRI_ATTRIB_RED_SIZE, &r);
        core->getConfigAttrib(configs[i], __DRI_ATTRIB_ALPHA_SIZE, &a);
        core->getConfigAttrib(configs[i], __DRI_ATTRIB_DEPTH_SIZE, &d);
        core->getConfigAttrib(configs[i], __DRI_ATTRIB_DOUBLE_BUFFER, &db);
        if (r == 8 && a == 8 && d == 24 && !db) { cfg = configs[i]; break; }
    }
    if (!cfg) return 3;
    unsigned err = 0;
    uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 3, __DRI_CTX_ATTRIB_MINOR_VERSION, 3};
    __DRIcontext *ctx = sw->createContextAttribs(scr, __DRI_API_OPENGL_CORE, cfg, NULL, 2, attribs, &err, NULL);
    if (!ctx) { fprintf(stderr, "no 3.3 core context (%u)\n", err); return 3; }
    __DRIdrawable *dr = sw->createNewDrawable(scr, cfg, NULL);
    if (!dr || !core->bindContext(ctx, dr, dr)) return 3;
    gpa = dlsym(RTLD_DEFAULT, "_glapi_get_proc_address");
    if (!gpa) return 2;

    GLF(PFNGLGENFRAMEBUFFERSPROC, glGenFramebuffers) GLF(PFNGLBINDFRAMEBUFFERPROC, glBindFramebuffer)
    GLF(PFNGLGENRENDERBUFFERSPROC, glGenRenderbuffers) GLF(PFNGLBINDRENDERBUFFERPROC, glBindRenderbuffer)
    GLF(PFNGLRENDERBUFFERSTORAGEPROC, glRenderbufferStorage) GLF(PFNGLFRAMEBUFFERRENDERBUFFERPROC, glFramebufferRenderbuffer)
    GLF(PFNGLCHECKFRAMEBUFFERSTATUSPROC, glCheckFramebufferStatus) GLF(PFNGLCREATESHADERPROC, glCreateShader)
    GLF(PFNGLSHADERSOURCEPROC, glShaderSource) GLF(PFNGLCOMPILESHADERPROC, glCompileShader) GLF(PFNGLGETSHADERIVPROC, glGetShaderiv)
    GLF(PFNGLGETSHADERINFOLOGPROC, glGetShaderInfoLog) GLF(PFNGLCREATEPROGRAMPROC, glCreateProgram)
    GLF(PFNGLATTACHSHADERPROC, glAttachShader) GLF(PFNGLLINKPROGRAMPROC, glLinkProgram) GLF(PFNGLGETPROGRAMIVPROC, glGetProgramiv)
    GLF(PFNGLGETPROGRAMINFOLOGPROC, glGetProgramInfoLog) GLF(PFNGLUSEPROGRAMPROC, glUseProgram)
    GLF(PFNGLGETUNIFORMLOCATIONPROC, glGetUniformLocation) GLF(PFNGLUNIFORMMATRIX4FVPROC, glUniformMatrix4fv)
    GLF(PFNGLUNIFORM1IPROC, glUniform1i) GLF(PFNGLGENBUFFERSPROC, glGenBuffers) GLF(PFNGLBINDBUFFERPROC, glBindBuffer)
    GLF(PFNGLBUFFERDATAPROC, glBufferData) GLF(PFNGLGENVERTEXARRAYSPROC, glGenVertexArrays)
    GLF(PFNGLBINDVERTEXARRAYPROC, glBindVertexArray) GLF(PFNGLENABLEVERTEXATTRIBARRAYPROC, glEnableVertexAttribArray)
    GLF(PFNGLVERTEXATTRIBPOINTERPROC, glVertexAttribPointer)
    typedef void (*V4)(GLint, GLint, GLsizei, GLsizei);
    typedef void (*C4)(GLfloat, GLfloat, GLfloat, GLfloat);
    V4 glViewport_ = (V4)gpa("glViewport");
    C4 glClearColor_ = (C4)gpa("glClearColor");
    void (*glClearDepth_)(GLdouble) = gpa("glClearDepth");
    void (*glClear_)(GLbitfield) = gpa("glClear");
    void (*glEnable_)(GLenum) = gpa("glEnable");
    void (*glGetFloatv_)(GLenum, GLfloat *) = gpa("glGetFloatv");
    void (*glLineWidth_)(GLfloat) = gpa("glLineWidth");
    void (*glDrawArrays_)(GLenum, GLint, GLsizei) = gpa("glDrawArrays");
    void (*glDrawElements_)(GLenum, GLsizei, GLenum, const void *) = gpa("glDrawElements");
    void (*glReadPixels_)(GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, void *) = gpa("glReadPixels");
    void (*glPixelStorei_)(GLenum, GLint) = gpa("glPixelStorei");
    void (*glFinish_)(void) = gpa("glFinish");
    GLenum (*glGetError_)(void) = gpa("glGetError");
    const GLubyte *(*glGetString_)(GLenum) = gpa("glGetString");

    /* the four shader sources */
    FILE *sf = fopen(argv[2], "rb");
    if (!sf) return 4;
    fseek(sf, 0, SEEK_END);
    long sl = ftell(sf);
    fseek(sf, 0, SEEK_SET);
    char *stext = calloc(1, (size_t)sl + 1);
    rd(stext, (size_t)sl, sf);
    fclose(sf);
    char *src[4] = {stext, NULL, NULL, NULL};
    for (int i = 1; i < 4; ++i) {
        char *sep = strstr(src[i - 1], "\n=====\n");
        if (!sep) { fprintf(stderr, "shaders.txt needs four sections\n"); return 4; }
        *sep = 0;
        src[i] = sep + 7;
    }
    GLuint prog[2];
    for (int p = 0; p < 2; ++p) {
        prog[p] = glCreateProgram();
        for (int s = 0; s < 2; ++s) {
            GLuint sh = glCreateShader(s == 0 ? GL_VERTEX_SHADER : GL_FRAGMENT_SHADER);
            const char *t = src[2 * p + s];
            glShaderSource(sh, 1, &t, NULL);
            glCompileShader(sh);
            GLint ok = 0;
            glGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
            if (!ok) {
                char log[4096];
                glGetShaderInfoLog(sh, sizeof log, NULL, log);
                fprintf(stderr, "shader %d/%d: %s\n", p, s, log);
                return 5;
            }
            glAttachShader(prog[p], sh);
        }
        glLinkProgram(prog[p]);
        GLint ok = 0;
        glGetProgramiv(prog[p], GL_LINK_STATUS, &ok);
        if (!ok) {
            char log[4096];
            glGetProgramInfoLog(prog[p], sizeof log, NULL, log);
            fprintf(stderr, "link %d: %s\n", p, log);
            return 5;
        }
    }

    FILE *f = fopen(argv[1], "rb");
    if (!f) return 4;
    int32_t W, H, depth_bits, ndraw;
    float clear[3];
    rd(&W, 4, f); rd(&H, 4, f); rd(clear, 12, f); rd(&depth_bits, 4, f); rd(&ndraw, 4, f);
    GLuint fbo, rb[2];
    glGenFramebuffers(1, &fbo);
    glBindFramebuffer(GL_FRAMEBUFFER, fbo);
    glGenRenderbuffers(2, rb);
    glBindRenderbuffer(GL_RENDERBUFFER, rb[0]);
    glRenderbufferStorage(GL_RENDERBUFFER, GL_RGB32F, W, H);          /* ctx.renderbuffer(components=3, dtype='f4') */
    glFramebufferRenderbuffer(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_RENDERBUFFER, rb[0]);
    glBindRenderbuffer(GL_RENDERBUFFER, rb[1]);
    glRenderbufferStorage(GL_RENDERBUFFER, depth_bits == 32 ? GL_DEPTH_COMPONENT32F : GL_DEPTH_COMPONENT24, W, H);
    glFramebufferRenderbuffer(GL_FRAMEBUFFER, GL_DEPTH_ATTACHMENT, GL_RENDERBUFFER, rb[1]);
    if (glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) { fprintf(stderr, "framebuffer incomplete\n"); return 6; }
    glViewport_(0, 0, W, H);
    glClearColor_(clear[0], clear[1], clear[2], 1.0f);
    glClearDepth_(1.0);
    glClear_(GL_COLOR_BUFFER_BIT | GL_DEPTH_BUFFER_BIT);
    glEnable_(GL_DEPTH_TEST);
    float lw_range[2] = {0, 0};
    glGetFloatv_(GL_ALIASED_LINE_WIDTH_RANGE, lw_range);
    fprintf(stderr, "%s / %s, line widths %g..%g\n", glGetString_(GL_VERSION), glGetString_(GL_RENDERER), lw_range[0], lw_range[1]);

    for (int k = 0; k < ndraw; ++k) {
        int32_t kind, nverts, nidx, option;
        float lw, proj[16], pose[16];
        rd(&kind, 4, f); rd(&nverts, 4, f); rd(&nidx, 4, f); rd(&option, 4, f); rd(&lw, 4, f);
        rd(proj, 64, f); rd(pose, 64, f);
        float *pos = malloc((size_t)nverts * 12), *tan = NULL;
        uint32_t *idx = NULL;
        rd(pos, (size_t)nverts * 12, f);
        if (kind == 1) { tan = malloc((size_t)nverts * 12); rd(tan, (size_t)nverts * 12, f); }
        else { idx = malloc((size_t)nidx * 4); rd(idx, (size_t)nidx * 4, f); }
        GLuint pr = prog[kind == 1], vao, bo[3];
        glUseProgram(pr);
        glUniformMatrix4fv(glGetUniformLocation(pr, "projection"), 1, GL_TRUE /* row major in memory */, proj);
        glUniformMatrix4fv(glGetUniformLocation(pr, "transform"), 1, GL_TRUE, pose);
        glUniform1i(glGetUniformLocation(pr, kind == 1 ? "colorOption" : "depthOption"), option);
        glGenVertexArrays(1, &vao);
        glBindVertexArray(vao);
        glGenBuffers(3, bo);
        glBindBuffer(GL_ARRAY_BUFFER, bo[0]);
        glBufferData(GL_ARRAY_BUFFER, (GLsizeiptr)nverts * 12, pos, GL_STATIC_DRAW);
        glEnableVertexAttribArray(0);
        glVertexAttribPointer(0, 3, GL_FLOAT, GL_FALSE, 0, 0);
        if (kind == 1) {
            glBindBuffer(GL_ARRAY_BUFFER, bo[1]);
            glBufferData(GL_ARRAY_BUFFER, (GLsizeiptr)nverts * 12, tan, GL_STATIC_DRAW);
            glEnableVertexAttribArray(1);
            glVertexAttribPointer(1, 3, GL_FLOAT, GL_FALSE, 0, 0);
            glLineWidth_(lw);
            glDrawArrays_(GL_LINES, 0, nverts);
        } else {
            glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, bo[2]);
            glBufferData(GL_ELEMENT_ARRAY_BUFFER, (GLsizeiptr)nidx * 4, idx, GL_STATIC_DRAW);
            glDrawElements_(GL_TRIANGLES, nidx, GL_UNSIGNED_INT, 0);
        }
        GLenum e = glGetError_();
        if (e) fprintf(stderr, "draw %d: GL error 0x%x\n", k, e);
        free(pos); free(tan); free(idx);
    }
    fclose(f);
    glFinish_();
    float *img = malloc((size_t)W * H * 16);
    glPixelStorei_(GL_PACK_ALIGNMENT, 1);
    glReadPixels_(0, 0, W, H, GL_RGBA, GL_FLOAT, img);
    GLenum e = glGetError_();
    if (e) { fprintf(stderr, "read: GL error 0x%x\n", e); return 7; }
    FILE *o = fopen(argv[3], "wb");
    fwrite(lw_range, 4, 2, o);
    fwrite(img, 16, (size_t)W * H, o);
    fclose(o);
    return 0;
}
