/* gl_ref.c -- test infrastructure, build container only: draws the two passes of Utils/Render_utils.py (triangles of
 * BustObj :130-200, GL_LINES of StrandsObj :8-127, Renderer :203-262) with a REAL OpenGL implementation and writes the
 * float colour buffer, so that csrc/raster.hip and oracle/raster_oracle.c can be pinned to a GL rasteriser
 * (tools/gen_golden_gl.py -> tests/golden/gl_*.npz).
 *
 * The GL is Google SwiftShader (OpenGL ES 3.0, software, headless through an EGL pbuffer), which this image happens to
 * ship inside the `kaleido` Python package; nothing here is installed or downloaded.  The GLSL below is a restatement
 * for ES 3.00 of what the reference's desktop-GLSL shaders compute (same uniforms, same arithmetic):
 *   vertex:   camera_v = transform * position;  gl_Position = projection * camera_v;  depth = -camera_v.z;
 *             lines also: the NDC displacement of a 0.01 step along the unit tangent (Tangent_2d)
 *   fragment: option 0: depth / 2 in r,g,b;  triangles 1: black, 2: white;
 *             lines 1: ((cos t, sin t, 0) + (1,1,0)) / 2, 2: the same with 2t, 3: white;  t = atan(Tangent_2d.y, Tangent_2d.x)
 * State as the reference sets it: RGBA32F colour + 24-bit depth renderbuffers, viewport = buffer size, clear colour from
 * the job, clear depth 1, DEPTH_TEST with the default LESS, no culling, line width from the job (moderngl: ctx.line_width).
 *
 *   gl_ref <libEGL.so> <libGLESv2.so> <job.bin> <out.bin>
 * job.bin (little endian): int32 W, H; float clear[3]; int32 depth_bits (24 | 32); int32 ndraw; then per draw
 *   int32 kind (0 triangles, 1 lines), nverts, nidx, option; float line_width; float projection[16], transform[16]
 *   (row major, as numpy holds them); float pos[nverts*3]; lines: float tangent[nverts*3]; triangles: uint32 idx[nidx]
 * out.bin: float32 line_width_range[2]; float32 RGBA [H][W], rows bottom-up as glReadPixels returns them.            */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef void *EGLDisplay, *EGLConfig, *EGLSurface, *EGLContext;
typedef int EGLint;
typedef unsigned int EGLBoolean, GLenum, GLuint, GLbitfield;
typedef int GLint, GLsizei;
typedef float GLfloat;
typedef unsigned char GLboolean;
typedef char GLchar;
typedef long GLsizeiptr;

#define F(ret, name, args) static ret(*name) args
F(EGLDisplay, eglGetDisplay, (void *));
F(EGLBoolean, eglInitialize, (EGLDisplay, EGLint *, EGLint *));
F(EGLBoolean, eglChooseConfig, (EGLDisplay, const EGLint *, EGLConfig *, EGLint, EGLint *));
F(EGLSurface, eglCreatePbufferSurface, (EGLDisplay, EGLConfig, const EGLint *));
F(EGLContext, eglCreateContext, (EGLDisplay, EGLConfig, EGLContext, const EGLint *));
F(EGLBoolean, eglMakeCurrent, (EGLDisplay, EGLSurface, EGLSurface, EGLContext));
F(EGLBoolean, eglBindAPI, (unsigned));
F(void, glGenFramebuffers, (GLsizei, GLuint *));
F(void, glBindFramebuffer, (GLenum, GLuint));
F(void, glGenRenderbuffers, (GLsizei, GLuint *));
F(void, glBindRenderbuffer, (GLenum, GLuint));
F(void, glRenderbufferStorage, (GLenum, GLenum, GLsizei, GLsizei));
F(void, glFramebufferRenderbuffer, (GLenum, GLenum, GLenum, GLuint));
F(GLenum, glCheckFramebufferStatus, (GLenum));
F(void, glViewport, (GLint, GLint, GLsizei, GLsizei));
F(void, glClearColor, (GLfloat, GLfloat, GLfloat, GLfloat));
F(void, glClearDepthf, (GLfloat));
F(void, glClear, (GLbitfield));
F(void, glEnable, (GLenum));
F(GLuint, glCreateShader, (GLenum));
F(void, glShaderSource, (GLuint, GLsizei, const GLchar *const *, const GLint *));
F(void, glCompileShader, (GLuint));
F(void, glGetShaderiv, (GLuint, GLenum, GLint *));
F(void, glGetShaderInfoLog, (GLuint, GLsizei, GLsizei *, GLchar *));
F(GLuint, glCreateProgram, (void));
F(void, glAttachShader, (GLuint, GLuint));
F(void, glLinkProgram, (GLuint));
F(void, glGetProgramiv, (GLuint, GLenum, GLint *));
F(void, glGetProgramInfoLog, (GLuint, GLsizei, GLsizei *, GLchar *));
F(void, glUseProgram, (GLuint));
F(GLint, glGetUniformLocation, (GLuint, const GLchar *));
F(void, glUniformMatrix4fv, (GLint, GLsizei, GLboolean, const GLfloat *));
F(void, glUniform1i, (GLint, GLint));
F(void, glGenBuffers, (GLsizei, GLuint *));
F(void, glBindBuffer, (GLenum, GLuint));
F(void, glBufferData, (GLenum, GLsizeiptr, const void *, GLenum));
F(void, glGenVertexArrays, (GLsizei, GLuint *));
F(void, glBindVertexArray, (GLuint));
F(void, glEnableVertexAttribArray, (GLuint));
F(void, glVertexAttribPointer, (GLuint, GLint, GLenum, GLboolean, GLsizei, const void *));
F(void, glDrawElements, (GLenum, GLsizei, GLenum, const void *));
F(void, glDrawArrays, (GLenum, GLint, GLsizei));
F(void, glLineWidth, (GLfloat));
F(void, glReadPixels, (GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, void *));
F(void, glFinish, (void));
F(GLenum, glGetError, (void));
F(void, glGetFloatv, (GLenum, GLfloat *));
F(void, glPixelStorei, (GLenum, GLint));

static void *need(void *lib, const char *n) {
    void *p = dlsym(lib, n);
    if (!p) {
        fprintf(stderr, "missing %s\n", n);
        exit(2);
    }
    return p;
}
#define L(lib, name) *(void **)&name = need(lib, #name)

static const char *VS_TRI =
    "#version 300 es\n"
    "uniform mat4 projection;\nuniform mat4 transform;\n"
    "layout(location = 0) in vec3 position;\n"
    "out highp float depth;\n"
    "void main() {\n"
    "  vec4 camera_v = transform * vec4(position, 1.0);\n"
    "  gl_Position = projection * camera_v;\n"
    "  depth = -camera_v.z;\n"
    "}\n";
static const char *FS_TRI =
    "#version 300 es\nprecision highp float;\nprecision highp int;\n"
    "uniform int option;\nin highp float depth;\nout vec4 colour;\n"
    "void main() {\n"
    "  if (option == 0) { float d = depth / 2.0; colour = vec4(d, d, d, 1.0); }\n"
    "  else if (option == 1) colour = vec4(0.0, 0.0, 0.0, 1.0);\n"
    "  else colour = vec4(1.0, 1.0, 1.0, 1.0);\n"
    "}\n";
static const char *VS_LINE =
    "#version 300 es\n"
    "uniform mat4 projection;\nuniform mat4 transform;\n"
    "layout(location = 0) in vec3 position;\nlayout(location = 1) in vec3 tangent;\n"
    "out highp float depth;\nout highp vec2 tangent_2d;\n"
    "void main() {\n"
    "  vec4 camera_v = transform * vec4(position, 1.0);\n"
    "  gl_Position = projection * camera_v;\n"
    "  vec2 here = gl_Position.xy / gl_Position.w;\n"
    "  vec3 ahead = position + normalize(tangent) * 0.01;\n"
    "  vec4 ahead_clip = projection * transform * vec4(ahead, 1.0);\n"
    "  tangent_2d = ahead_clip.xy / ahead_clip.w - here;\n"
    "  depth = -camera_v.z;\n"
    "}\n";
static const char *FS_LINE =
    "#version 300 es\nprecision highp float;\nprecision highp int;\n"
    "uniform int option;\nin highp float depth;\nin highp vec2 tangent_2d;\nout vec4 colour;\n"
    "void main() {\n"
    "  float t = atan(tangent_2d.y, tangent_2d.x);\n"
    "  if (option == 0) { float d = depth / 2.0; colour = vec4(d, d, d, 1.0); }\n"
    "  else if (option == 1) colour = vec4((vec3(cos(t), sin(t), 0.0) + vec3(1.0, 1.0, 0.0)) * 0.5, 1.0);\n"
    "  else if (option == 2) colour = vec4((vec3(cos(2.0 * t), sin(2.0 * t), 0.0) + vec3(1.0, 1.0, 0.0)) * 0.5, 1.0);\n"
    "  else colour = vec4(1.0, 1.0, 1.0, 1.0);\n"
    "}\n";

static GLuint shader(GLenum type, const char *src) {
    GLuint s = glCreateShader(type);
    glShaderSource(s, 1, &src, NULL);
    glCompileShader(s);
    GLint ok = 0;
    glGetShaderiv(s, 0x8B81, &ok);
    if (!ok) {
        char log[4096];
        glGetShaderInfoLog(s, sizeof log, NULL, log);
        fprintf(stderr, "shader: %s\n", log);
        exit(3);
    }
    return s;
}
static GLuint program(const char *vs, const char *fs) {
    GLuint p = glCreateProgram();
    glAttachShader(p, shader(0x8B31, vs));
    glAttachShader(p, shader(0x8B30, fs));
    glLinkProgram(p);
    GLint ok = 0;
    glGetProgramiv(p, 0x8B82, &ok);
    if (!ok) {
        char log[4096];
        glGetProgramInfoLog(p, sizeof log, NULL, log);
        fprintf(stderr, "link: %s\n", log);
        exit(3);
    }
    return p;
}
static void rd(void *dst, size_t n, FILE *f) {
    if (fread(dst, 1, n, f) != n) {
        fprintf(stderr, "short job file\n");
        exit(4);
    }
}

int main(int argc, char **argv) {
    if (argc < 5) return 1;
    void *egl = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL), *gl = dlopen(argv[2], RTLD_NOW | RTLD_GLOBAL);
    if (!egl || !gl) {
        fprintf(stderr, "dlopen: %s\n", dlerror());
        return 2;
    }
    L(egl, eglGetDisplay); L(egl, eglInitialize); L(egl, eglChooseConfig); L(egl, eglCreatePbufferSurface);
    L(egl, eglCreateContext); L(egl, eglMakeCurrent); L(egl, eglBindAPI);
    L(gl, glGenFramebuffers); L(gl, glBindFramebuffer); L(gl, glGenRenderbuffers); L(gl, glBindRenderbuffer);
    L(gl, glRenderbufferStorage); L(gl, glFramebufferRenderbuffer); L(gl, glCheckFramebufferStatus); L(gl, glViewport);
    L(gl, glClearColor); L(gl, glClearDepthf); L(gl, glClear); L(gl, glEnable); L(gl, glCreateShader);
    L(gl, glShaderSource); L(gl, glCompileShader); L(gl, glGetShaderiv); L(gl, glGetShaderInfoLog);
    L(gl, glCreateProgram); L(gl, glAttachShader); L(gl, glLinkProgram); L(gl, glGetProgramiv);
    L(gl, glGetProgramInfoLog); L(gl, glUseProgram); L(gl, glGetUniformLocation); L(gl, glUniformMatrix4fv);
    L(gl, glUniform1i); L(gl, glGenBuffers); L(gl, glBindBuffer); L(gl, glBufferData); L(gl, glGenVertexArrays);
    L(gl, glBindVertexArray); L(gl, glEnableVertexAttribArray); L(gl, glVertexAttribPointer); L(gl, glDrawElements);
    L(gl, glDrawArrays); L(gl, glLineWidth); L(gl, glReadPixels); L(gl, glFinish); L(gl, glGetError);
    L(gl, glGetFloatv); L(gl, glPixelStorei);

    FILE *f = fopen(argv[3], "rb");
    if (!f) return 4;
    int32_t W, H, depth_bits, ndraw;
    float clear[3];
    rd(&W, 4, f); rd(&H, 4, f); rd(clear, 12, f); rd(&depth_bits, 4, f); rd(&ndraw, 4, f);

    EGLDisplay d = eglGetDisplay(0);
    EGLint ma, mi;
    if (!eglInitialize(d, &ma, &mi)) return 5;
    const EGLint ca[] = {0x3033, 0x0001, 0x3040, 0x0040, 0x3024, 8, 0x3023, 8, 0x3022, 8, 0x3025, 24, 0x3038};
    EGLConfig cfg;
    EGLint n;
    if (!eglChooseConfig(d, ca, &cfg, 1, &n) || n < 1) return 5;
    const EGLint pa[] = {0x3057, 16, 0x3056, 16, 0x3038};
    EGLSurface s = eglCreatePbufferSurface(d, cfg, pa);
    eglBindAPI(0x30A0);
    const EGLint cx[] = {0x3098, 3, 0x3038};
    EGLContext c = eglCreateContext(d, cfg, 0, cx);
    if (!c || !eglMakeCurrent(d, s, s, c)) return 5;

    GLuint fbo, rb[2];
    glGenFramebuffers(1, &fbo);
    glBindFramebuffer(0x8D40, fbo);
    glGenRenderbuffers(2, rb);
    glBindRenderbuffer(0x8D41, rb[0]);
    glRenderbufferStorage(0x8D41, 0x8814 /* RGBA32F */, W, H);
    glFramebufferRenderbuffer(0x8D40, 0x8CE0, 0x8D41, rb[0]);
    glBindRenderbuffer(0x8D41, rb[1]);
    glRenderbufferStorage(0x8D41, depth_bits == 32 ? 0x8CAC /* DEPTH_COMPONENT32F */ : 0x81A6 /* DEPTH_COMPONENT24 */, W, H);
    glFramebufferRenderbuffer(0x8D40, 0x8D00, 0x8D41, rb[1]);
    if (glCheckFramebufferStatus(0x8D40) != 0x8CD5) {
        fprintf(stderr, "framebuffer incomplete\n");
        return 6;
    }
    glViewport(0, 0, W, H);
    glClearColor(clear[0], clear[1], clear[2], 1.0f);
    glClearDepthf(1.0f);
    glClear(0x4000 | 0x0100);
    glEnable(0x0B71 /* DEPTH_TEST, func LESS by default */);
    float lw_range[2] = {0, 0};
    glGetFloatv(0x846E /* ALIASED_LINE_WIDTH_RANGE */, lw_range);

    GLuint ptri = program(VS_TRI, FS_TRI), pline = program(VS_LINE, FS_LINE);
    for (int k = 0; k < ndraw; ++k) {
        int32_t kind, nverts, nidx, option;
        float lw, proj[16], pose[16];
        rd(&kind, 4, f); rd(&nverts, 4, f); rd(&nidx, 4, f); rd(&option, 4, f); rd(&lw, 4, f);
        rd(proj, 64, f); rd(pose, 64, f);
        float *pos = malloc((size_t)nverts * 12), *tan = NULL;
        uint32_t *idx = NULL;
        rd(pos, (size_t)nverts * 12, f);
        if (kind == 1) {
            tan = malloc((size_t)nverts * 12);
            rd(tan, (size_t)nverts * 12, f);
        } else {
            idx = malloc((size_t)nidx * 4);
            rd(idx, (size_t)nidx * 4, f);
        }
        GLuint prog = kind == 1 ? pline : ptri, vao, bo[3];
        glUseProgram(prog);
        glUniformMatrix4fv(glGetUniformLocation(prog, "projection"), 1, 1 /* row major in memory */, proj);
        glUniformMatrix4fv(glGetUniformLocation(prog, "transform"), 1, 1, pose);
        glUniform1i(glGetUniformLocation(prog, "option"), option);
        glGenVertexArrays(1, &vao);
        glBindVertexArray(vao);
        glGenBuffers(3, bo);
        glBindBuffer(0x8892, bo[0]);
        glBufferData(0x8892, (GLsizeiptr)nverts * 12, pos, 0x88E4);
        glEnableVertexAttribArray(0);
        glVertexAttribPointer(0, 3, 0x1406, 0, 0, 0);
        if (kind == 1) {
            glBindBuffer(0x8892, bo[1]);
            glBufferData(0x8892, (GLsizeiptr)nverts * 12, tan, 0x88E4);
            glEnableVertexAttribArray(1);
            glVertexAttribPointer(1, 3, 0x1406, 0, 0, 0);
            glLineWidth(lw);
            glDrawArrays(0x0001 /* LINES */, 0, nverts);
        } else {
            glBindBuffer(0x8893, bo[2]);
            glBufferData(0x8893, (GLsizeiptr)nidx * 4, idx, 0x88E4);
            glDrawElements(0x0004 /* TRIANGLES */, nidx, 0x1405 /* UNSIGNED_INT */, 0);
        }
        GLenum e = glGetError();
        if (e) fprintf(stderr, "draw %d: GL error 0x%x\n", k, e);
        free(pos); free(tan); free(idx);
    }
    fclose(f);
    glFinish();
    float *img = malloc((size_t)W * H * 16);
    glPixelStorei(0x0D05 /* PACK_ALIGNMENT */, 1);
    glReadPixels(0, 0, W, H, 0x1908 /* RGBA */, 0x1406 /* FLOAT */, img);
    GLenum e = glGetError();
    if (e) {
        fprintf(stderr, "read: GL error 0x%x\n", e);
        return 7;
    }
    FILE *o = fopen(argv[4], "wb");
    fwrite(lw_range, 4, 2, o);
    fwrite(img, 16, (size_t)W * H, o);
    fclose(o);
    return 0;
}
