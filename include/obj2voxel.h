/*
 * obj2voxel.h -- public C API of the MI355X-native voxelizer.
 *
 * Drop-in for the reference's include/obj2voxel.h: the same 35 extern "C" entry points, typedefs, callback
 * shapes, enum values and error codes (reference include/obj2voxel.h:15-79 for the types and constants,
 * :89-406 for the functions; each declaration below cites the line it replaces).  A program written against
 * the reference header recompiles and links against libobj2voxel_amd.so unchanged.
 *
 * What differs behind the API: obj2voxel_voxelize() runs the per-triangle voxelization on one MI355X through
 * the C-ABI in o2v_hip.h instead of the CPU chunk loop and worker pool.  The worker entry points still exist
 * and keep their contract (run_worker blocks until stop_workers) but no work is dispatched to them.
 */
#ifndef OBJ2VOXEL_HEADER
#define OBJ2VOXEL_HEADER

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- types (reference :15-38) ------------------------------------------------------------------------- */

typedef unsigned char obj2voxel_enum_t;  /* enum-like constants */
typedef unsigned char obj2voxel_byte_t;  /* raw bytes */
typedef unsigned char obj2voxel_error_t; /* result of obj2voxel_voxelize */

/* Opaque. Layouts are implementation-defined in the reference as well (:22-28). */
typedef struct obj2voxel_instance obj2voxel_instance;
typedef struct obj2voxel_texture obj2voxel_texture;
typedef struct obj2voxel_triangle obj2voxel_triangle;

/* Pulled until it returns false; fill out_triangle through obj2voxel_set_triangle_*. (:32) */
typedef bool(obj2voxel_triangle_callback)(void *callback_data, obj2voxel_triangle *out_triangle);
/* Receives voxel_count records of four host-endian uint32: x, y, z, argb. Return false on failure. (:35) */
typedef bool(obj2voxel_voxel_callback)(void *callback_data, uint32_t *voxel_data, size_t voxel_count);
/* Return true if the message was consumed, false to have it printed. (:38) */
typedef bool(obj2voxel_log_callback)(void *callback_data, const char *msg, obj2voxel_enum_t level);

/* ---- constants (reference :43-79) ---------------------------------------------------------------------- */

static const obj2voxel_enum_t OBJ2VOXEL_MAX_STRATEGY = 0;   /* colour of the heaviest triangle wins */
static const obj2voxel_enum_t OBJ2VOXEL_BLEND_STRATEGY = 1; /* weighted average of triangle colours */

static const obj2voxel_enum_t OBJ2VOXEL_UV_CLAMP = 0;
static const obj2voxel_enum_t OBJ2VOXEL_UV_WRAP = 1;

static const obj2voxel_enum_t OBJ2VOXEL_LOG_LEVEL_SILENT = 0;
static const obj2voxel_enum_t OBJ2VOXEL_LOG_LEVEL_ERROR = 1;
static const obj2voxel_enum_t OBJ2VOXEL_LOG_LEVEL_WARNING = 2;
static const obj2voxel_enum_t OBJ2VOXEL_LOG_LEVEL_INFO = 3;
static const obj2voxel_enum_t OBJ2VOXEL_LOG_LEVEL_DEBUG = 4;

static const obj2voxel_error_t OBJ2VOXEL_ERR_OK = 0;
static const obj2voxel_error_t OBJ2VOXEL_ERR_NO_INPUT = 1;
static const obj2voxel_error_t OBJ2VOXEL_ERR_NO_OUTPUT = 2;
static const obj2voxel_error_t OBJ2VOXEL_ERR_NO_RESOLUTION = 3;
static const obj2voxel_error_t OBJ2VOXEL_ERR_IO_ERROR_ON_OPEN_INPUT_FILE = 4;
static const obj2voxel_error_t OBJ2VOXEL_ERR_IO_ERROR_ON_OPEN_OUTPUT_FILE = 5;
static const obj2voxel_error_t OBJ2VOXEL_ERR_IO_ERROR_DURING_VOXEL_WRITE = 6;
static const obj2voxel_error_t OBJ2VOXEL_ERR_DOUBLE_VOXELIZATION = 7; /* instances are single-use */
/* EXTENSION of this build (not in the reference, whose codes end at 7): no usable GPU, a HIP failure, out of device memory,
 * or a limit of the dense-grid path (sample resolution above 65 535, ...).  The reason is logged at ERROR level.  There is no
 * CPU voxelization path to fall back to. */
static const obj2voxel_error_t OBJ2VOXEL_ERR_DEVICE = 8;

/* ---- instance (reference :89-95) ----------------------------------------------------------------------- */

obj2voxel_instance *obj2voxel_alloc(void);
void obj2voxel_free(obj2voxel_instance *instance);

/* ---- logging, process-global (reference :105-120) ------------------------------------------------------ */

void obj2voxel_set_log_level(obj2voxel_enum_t level);
void obj2voxel_set_log_callback(obj2voxel_log_callback *callback, void *callback_data);
obj2voxel_enum_t obj2voxel_get_log_level(void);

/* ---- settings (reference :130-264) --------------------------------------------------------------------- */

void obj2voxel_set_resolution(obj2voxel_instance *instance, uint32_t resolution);   /* :130, non-zero */
void obj2voxel_set_supersampling(obj2voxel_instance *instance, uint32_t level);     /* :138, 1 or 2 */
void obj2voxel_set_color_strategy(obj2voxel_instance *instance, obj2voxel_enum_t strategy); /* :146 */
/* Fallback texture for file inputs; borrowed, must outlive voxelization. (:157) */
void obj2voxel_set_texture(obj2voxel_instance *instance, obj2voxel_texture *texture);
/* type: extension without dot ("obj", "stl") or NULL to detect from the path. Opened at voxelize time. (:167) */
void obj2voxel_set_input_file(obj2voxel_instance *instance, const char *file, const char *type);
void obj2voxel_set_input_callback(obj2voxel_instance *instance, obj2voxel_triangle_callback *callback,
                                  void *callback_data); /* :177 */
void obj2voxel_set_output_file(obj2voxel_instance *instance, const char *file, const char *type); /* :189 */
/* Keep the encoded output in memory; fetch it with obj2voxel_get_output_memory. (:198) */
void obj2voxel_set_output_memory(obj2voxel_instance *instance, const char *type);
void obj2voxel_set_output_callback(obj2voxel_instance *instance, obj2voxel_voxel_callback *callback,
                                   void *callback_data); /* :207 */
/* Kept for compatibility: the GPU path never dispatches to CPU workers. (:219) */
void obj2voxel_set_parallel(obj2voxel_instance *instance, bool enabled);
/* Row-major 3x3 of -1/0/1 that permutes / flips axes. (:229) */
void obj2voxel_set_unit_transform(obj2voxel_instance *instance, const int transform[9]);
/* min xyz then max xyz; skips the bounds reduce. (:238) */
void obj2voxel_set_mesh_boundaries(obj2voxel_instance *instance, const float bounds[6]);

uint32_t obj2voxel_get_resolution(obj2voxel_instance *instance); /* :246 */
uint32_t obj2voxel_get_chunk_size(obj2voxel_instance *instance); /* :255, always 64 */
/* NULL (out_size untouched) unless the output is a memory output. Valid until obj2voxel_free. (:264) */
const obj2voxel_byte_t *obj2voxel_get_output_memory(obj2voxel_instance *instance, size_t *out_size);

/* ---- triangles, to be called from the triangle callback (reference :273-294) ---------------------------- */

void obj2voxel_set_triangle_basic(obj2voxel_triangle *triangle, const float vertices[9]);
/* As in the reference the colour is stored but the triangle stays material-less (renders white). */
void obj2voxel_set_triangle_colored(obj2voxel_triangle *triangle, const float vertices[9], const float color[3]);
void obj2voxel_set_triangle_textured(obj2voxel_triangle *triangle, const float vertices[9], const float textures[6],
                                     obj2voxel_texture *texture);

/* ---- textures (reference :302-364) --------------------------------------------------------------------- */

obj2voxel_texture *obj2voxel_texture_alloc(void);
void obj2voxel_texture_free(obj2voxel_texture *texture);
bool obj2voxel_texture_load_from_file(obj2voxel_texture *texture, const char *file, const char *type);
bool obj2voxel_texture_load_from_memory(obj2voxel_texture *texture, const obj2voxel_byte_t *data, size_t size,
                                        const char *type);
/* 8-bit channels; channels == 3 is RGB, 4 is ARGB. The pixels are copied. (:334) */
bool obj2voxel_texture_load_pixels(obj2voxel_texture *texture, const obj2voxel_byte_t *pixels, size_t width,
                                   size_t height, size_t channels);
/* Spelled as in the reference (:350). */
void obj2voxel_teture_set_uv_mode(obj2voxel_texture *texture, obj2voxel_enum_t mode);
void obj2voxel_texture_get_meta(obj2voxel_texture *texture, size_t *out_width, size_t *out_height,
                                size_t *out_channels);
void obj2voxel_texture_get_pixels(obj2voxel_texture *texture, obj2voxel_byte_t *out_pixels);

/* ---- threading (reference :374-396) -------------------------------------------------------------------- */

/* Registers the calling thread and blocks until obj2voxel_stop_workers. */
void obj2voxel_run_worker(obj2voxel_instance *instance);
void obj2voxel_stop_workers(obj2voxel_instance *instance);
uint32_t obj2voxel_get_worker_count(obj2voxel_instance *instance);

/* ---- voxelization (reference :406) --------------------------------------------------------------------- */

obj2voxel_error_t obj2voxel_voxelize(obj2voxel_instance *instance);

#ifdef __cplusplus
}
#endif

#endif
