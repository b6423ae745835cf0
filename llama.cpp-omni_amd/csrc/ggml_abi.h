// ggml_abi.h -- the slice of the ggml binary interface the MI355X backend touches, restated.
//
// The backend is loaded by the reference through `ggml_backend_load()` / $GGML_BACKEND_PATH
// (reference: ggml/src/ggml-backend-reg.cpp:265-285, :603-607) and talks to it only through the
// plug-in structs of ggml/src/ggml-backend-impl.h:17-210 (GGML_BACKEND_API_VERSION 2) and the
// tensor / graph structs of ggml/include/ggml.h:626-658 and ggml/src/ggml-impl.h:327-341.
// This header declares exactly that layout so the product builds with no reference checkout.
// Every size / offset / enum value below is checked against the real headers by
// tests/test_abi.py (golden: tests/golden/abi.json, produced by oracle/abi_probe.c).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

// ---------------------------------------------------------------- limits (ggml.h:218-225)
#define GGML_MAX_DIMS       4
#define GGML_MAX_SRC        10
#define GGML_MAX_OP_PARAMS  64
#define GGML_MAX_NAME       64
#define GGML_KQ_MASK_PAD    64          // ggml.h:2177
#define GGML_BACKEND_API_VERSION 2      // ggml-backend-impl.h:11

// ---------------------------------------------------------------- enums (values, not order, matter)
enum ggml_status {                      // ggml.h:348-353
    GGML_STATUS_ALLOC_FAILED = -2,
    GGML_STATUS_FAILED       = -1,
    GGML_STATUS_SUCCESS      =  0,
    GGML_STATUS_ABORTED      =  1,
};

enum ggml_type {                        // ggml.h:379-421 (only the ones the backend names)
    GGML_TYPE_F32  = 0,  GGML_TYPE_F16  = 1,  GGML_TYPE_Q4_0 = 2,  GGML_TYPE_Q4_1 = 3,
    GGML_TYPE_Q5_0 = 6,  GGML_TYPE_Q5_1 = 7,  GGML_TYPE_Q8_0 = 8,  GGML_TYPE_Q8_1 = 9,
    GGML_TYPE_Q2_K = 10, GGML_TYPE_Q3_K = 11, GGML_TYPE_Q4_K = 12, GGML_TYPE_Q5_K = 13,
    GGML_TYPE_Q6_K = 14, GGML_TYPE_Q8_K = 15,
    GGML_TYPE_I8   = 24, GGML_TYPE_I16  = 25, GGML_TYPE_I32  = 26, GGML_TYPE_I64  = 27,
    GGML_TYPE_F64  = 28, GGML_TYPE_BF16 = 30, GGML_TYPE_MXFP4 = 39,
    GGML_TYPE_COUNT = 40,
};

enum ggml_prec { GGML_PREC_DEFAULT = 0, GGML_PREC_F32 = 10 };   // ggml.h:424-427

enum ggml_op {                          // ggml.h:459-566
    GGML_OP_NONE = 0, GGML_OP_DUP = 1, GGML_OP_ADD = 2, GGML_OP_ADD_ID = 3, GGML_OP_ADD1 = 4,
    GGML_OP_ACC = 5, GGML_OP_SUB = 6, GGML_OP_MUL = 7, GGML_OP_DIV = 8, GGML_OP_SQR = 9,
    GGML_OP_SQRT = 10, GGML_OP_LOG = 11, GGML_OP_SIN = 12, GGML_OP_COS = 13, GGML_OP_SUM = 14,
    GGML_OP_SUM_ROWS = 15, GGML_OP_MEAN = 16, GGML_OP_ARGMAX = 17, GGML_OP_COUNT_EQUAL = 18,
    GGML_OP_REPEAT = 19, GGML_OP_REPEAT_BACK = 20, GGML_OP_CONCAT = 21, GGML_OP_SILU_BACK = 22,
    GGML_OP_NORM = 23, GGML_OP_RMS_NORM = 24, GGML_OP_RMS_NORM_BACK = 25, GGML_OP_GROUP_NORM = 26,
    GGML_OP_L2_NORM = 27, GGML_OP_MUL_MAT = 28, GGML_OP_MUL_MAT_ID = 29, GGML_OP_OUT_PROD = 30,
    GGML_OP_SCALE = 31, GGML_OP_SET = 32, GGML_OP_CPY = 33, GGML_OP_CONT = 34, GGML_OP_RESHAPE = 35,
    GGML_OP_VIEW = 36, GGML_OP_PERMUTE = 37, GGML_OP_TRANSPOSE = 38, GGML_OP_GET_ROWS = 39,
    GGML_OP_GET_ROWS_BACK = 40, GGML_OP_SET_ROWS = 41, GGML_OP_DIAG = 42, GGML_OP_DIAG_MASK_INF = 43,
    GGML_OP_DIAG_MASK_ZERO = 44, GGML_OP_SOFT_MAX = 45, GGML_OP_SOFT_MAX_BACK = 46, GGML_OP_ROPE = 47,
    GGML_OP_ROPE_BACK = 48, GGML_OP_CLAMP = 49, GGML_OP_CONV_TRANSPOSE_1D = 50, GGML_OP_IM2COL = 51,
    GGML_OP_IM2COL_BACK = 52, GGML_OP_IM2COL_3D = 53, GGML_OP_CONV_2D = 54, GGML_OP_CONV_3D = 55,
    GGML_OP_CONV_2D_DW = 56, GGML_OP_CONV_TRANSPOSE_2D = 57, GGML_OP_POOL_1D = 58, GGML_OP_POOL_2D = 59,
    GGML_OP_POOL_2D_BACK = 60, GGML_OP_UPSCALE = 61, GGML_OP_PAD = 62, GGML_OP_PAD_REFLECT_1D = 63,
    GGML_OP_ROLL = 64, GGML_OP_ARANGE = 65, GGML_OP_TIMESTEP_EMBEDDING = 66, GGML_OP_ARGSORT = 67,
    GGML_OP_LEAKY_RELU = 68, GGML_OP_FLASH_ATTN_EXT = 69,
    GGML_OP_UNARY = 80,
    GGML_OP_GLU = 89,
    GGML_OP_COUNT = 90,
};
enum ggml_op_pool { GGML_OP_POOL_MAX = 0, GGML_OP_POOL_AVG = 1, GGML_OP_POOL_COUNT = 2 };      /* ggml.h:2024-2028 */

enum ggml_unary_op {                    // ggml.h:568-587
    GGML_UNARY_OP_ABS = 0, GGML_UNARY_OP_SGN = 1, GGML_UNARY_OP_NEG = 2, GGML_UNARY_OP_STEP = 3,
    GGML_UNARY_OP_TANH = 4, GGML_UNARY_OP_ELU = 5, GGML_UNARY_OP_RELU = 6, GGML_UNARY_OP_SIGMOID = 7,
    GGML_UNARY_OP_GELU = 8, GGML_UNARY_OP_GELU_QUICK = 9, GGML_UNARY_OP_SILU = 10,
    GGML_UNARY_OP_HARDSWISH = 11, GGML_UNARY_OP_HARDSIGMOID = 12, GGML_UNARY_OP_EXP = 13,
    GGML_UNARY_OP_GELU_ERF = 14, GGML_UNARY_OP_XIELU = 15, GGML_UNARY_OP_COUNT = 16,
};

enum ggml_glu_op {                      // ggml.h:589-598
    GGML_GLU_OP_REGLU = 0, GGML_GLU_OP_GEGLU = 1, GGML_GLU_OP_SWIGLU = 2, GGML_GLU_OP_SWIGLU_OAI = 3,
    GGML_GLU_OP_GEGLU_ERF = 4, GGML_GLU_OP_GEGLU_QUICK = 5,
};

enum ggml_log_level {                   // ggml.h:606-613
    GGML_LOG_LEVEL_NONE = 0, GGML_LOG_LEVEL_DEBUG = 1, GGML_LOG_LEVEL_INFO = 2,
    GGML_LOG_LEVEL_WARN = 3, GGML_LOG_LEVEL_ERROR = 4, GGML_LOG_LEVEL_CONT = 5,
};

enum ggml_tensor_flag {                 // ggml.h:616-621
    GGML_TENSOR_FLAG_INPUT = 1, GGML_TENSOR_FLAG_OUTPUT = 2, GGML_TENSOR_FLAG_PARAM = 4, GGML_TENSOR_FLAG_LOSS = 8,
};

#define GGML_ROPE_TYPE_NORMAL 0         // ggml.h:241-244
#define GGML_ROPE_TYPE_NEOX   2
#define GGML_ROPE_TYPE_MROPE  8
#define GGML_ROPE_TYPE_VISION 24

// ---------------------------------------------------------------- tensor (336 B; ggml.h:626-658)
struct ggml_backend_buffer;
struct ggml_tensor {
    enum ggml_type               type;                          //   0
    struct ggml_backend_buffer * buffer;                        //   8
    int64_t                      ne[GGML_MAX_DIMS];             //  16  elements per dim
    size_t                       nb[GGML_MAX_DIMS];             //  48  byte strides
    enum ggml_op                 op;                            //  80
    int32_t                      op_params[GGML_MAX_OP_PARAMS / sizeof(int32_t)];   // 84
    int32_t                      flags;                         // 148
    struct ggml_tensor *         src[GGML_MAX_SRC];             // 152
    struct ggml_tensor *         view_src;                      // 232
    size_t                       view_offs;                     // 240
    void *                       data;                          // 248
    char                         name[GGML_MAX_NAME];           // 256
    void *                       extra;                         // 320
    char                         padding[8];                    // 328
};

// ---------------------------------------------------------------- graph (private in the reference: ggml-impl.h:327-341)
struct ggml_hash_set { size_t size; uint32_t * used; struct ggml_tensor ** keys; };   // ggml-impl.h:224-228
struct ggml_cgraph {
    int                   size;
    int                   n_nodes;
    int                   n_leafs;
    struct ggml_tensor ** nodes;
    struct ggml_tensor ** grads;
    struct ggml_tensor ** grad_accs;
    struct ggml_tensor ** leafs;
    int32_t *             use_counts;
    struct ggml_hash_set  visited_hash_set;
    int                   order;
};

// ---------------------------------------------------------------- plug-in handles (ggml-backend.h:24-30)
typedef struct ggml_backend_buffer_type * ggml_backend_buffer_type_t;
typedef struct ggml_backend_buffer *      ggml_backend_buffer_t;
typedef struct ggml_backend_event *       ggml_backend_event_t;
typedef struct ggml_backend *             ggml_backend_t;
typedef void *                            ggml_backend_graph_plan_t;
typedef struct ggml_backend_reg *         ggml_backend_reg_t;
typedef struct ggml_backend_device *      ggml_backend_dev_t;
typedef uint8_t                           ggml_guid[16];        // ggml.h:673-674
typedef ggml_guid *                       ggml_guid_t;

enum ggml_backend_buffer_usage {        // ggml-backend.h:49-53
    GGML_BACKEND_BUFFER_USAGE_ANY = 0, GGML_BACKEND_BUFFER_USAGE_WEIGHTS = 1, GGML_BACKEND_BUFFER_USAGE_COMPUTE = 2,
};
enum ggml_backend_dev_type {            // ggml-backend.h:130-139
    GGML_BACKEND_DEVICE_TYPE_CPU = 0, GGML_BACKEND_DEVICE_TYPE_GPU = 1,
    GGML_BACKEND_DEVICE_TYPE_IGPU = 2, GGML_BACKEND_DEVICE_TYPE_ACCEL = 3,
};
struct ggml_backend_dev_caps {          // ggml-backend.h:142-151
    bool async, host_buffer, buffer_from_host_ptr, events;
};
struct ggml_backend_dev_props {         // ggml-backend.h:154-171
    const char *                 name;
    const char *                 description;
    size_t                       memory_free;
    size_t                       memory_total;
    enum ggml_backend_dev_type   type;
    const char *                 device_id;
    struct ggml_backend_dev_caps caps;
};

// ---------------------------------------------------------------- vtables (ggml-backend-impl.h)
struct ggml_backend_buffer_type_i {     // :17-29
    const char *          (*get_name)      (ggml_backend_buffer_type_t);
    ggml_backend_buffer_t (*alloc_buffer)  (ggml_backend_buffer_type_t, size_t size);
    size_t                (*get_alignment) (ggml_backend_buffer_type_t);
    size_t                (*get_max_size)  (ggml_backend_buffer_type_t);
    size_t                (*get_alloc_size)(ggml_backend_buffer_type_t, const struct ggml_tensor *);
    bool                  (*is_host)       (ggml_backend_buffer_type_t);
};
struct ggml_backend_buffer_type {       // :31-35
    struct ggml_backend_buffer_type_i iface;
    ggml_backend_dev_t                device;
    void *                            context;
};

struct ggml_backend_buffer_i {          // :41-58
    void             (*free_buffer)  (ggml_backend_buffer_t);
    void *           (*get_base)     (ggml_backend_buffer_t);
    enum ggml_status (*init_tensor)  (ggml_backend_buffer_t, struct ggml_tensor *);
    void             (*memset_tensor)(ggml_backend_buffer_t, struct ggml_tensor *, uint8_t value, size_t offset, size_t size);
    void             (*set_tensor)   (ggml_backend_buffer_t, struct ggml_tensor *, const void * data, size_t offset, size_t size);
    void             (*get_tensor)   (ggml_backend_buffer_t, const struct ggml_tensor *, void * data, size_t offset, size_t size);
    bool             (*cpy_tensor)   (ggml_backend_buffer_t, const struct ggml_tensor * src, struct ggml_tensor * dst);
    void             (*clear)        (ggml_backend_buffer_t, uint8_t value);
    void             (*reset)        (ggml_backend_buffer_t);
};
struct ggml_backend_buffer {            // :60-66
    struct ggml_backend_buffer_i   iface;
    ggml_backend_buffer_type_t     buft;
    void *                         context;
    size_t                         size;
    enum ggml_backend_buffer_usage usage;
};

struct ggml_backend_i {                 // :87-120
    const char *     (*get_name)        (ggml_backend_t);
    void             (*free)            (ggml_backend_t);
    void             (*set_tensor_async)(ggml_backend_t, struct ggml_tensor *, const void * data, size_t offset, size_t size);
    void             (*get_tensor_async)(ggml_backend_t, const struct ggml_tensor *, void * data, size_t offset, size_t size);
    bool             (*cpy_tensor_async)(ggml_backend_t src_backend, ggml_backend_t dst_backend, const struct ggml_tensor * src, struct ggml_tensor * dst);
    void             (*synchronize)     (ggml_backend_t);
    ggml_backend_graph_plan_t (*graph_plan_create) (ggml_backend_t, const struct ggml_cgraph *);
    void             (*graph_plan_free)   (ggml_backend_t, ggml_backend_graph_plan_t);
    void             (*graph_plan_update) (ggml_backend_t, ggml_backend_graph_plan_t, const struct ggml_cgraph *);
    enum ggml_status (*graph_plan_compute)(ggml_backend_t, ggml_backend_graph_plan_t);
    enum ggml_status (*graph_compute)   (ggml_backend_t, struct ggml_cgraph *);
    void             (*event_record)    (ggml_backend_t, ggml_backend_event_t);
    void             (*event_wait)      (ggml_backend_t, ggml_backend_event_t);
    void             (*graph_optimize)  (ggml_backend_t, struct ggml_cgraph *);
};
struct ggml_backend {                   // :122-127
    ggml_guid_t           guid;
    struct ggml_backend_i iface;
    ggml_backend_dev_t    device;
    void *                context;
};
struct ggml_backend_event {             // :129-132
    struct ggml_backend_device * device;
    void *                       context;
};

struct ggml_backend_device_i {          // :140-182
    const char *               (*get_name)            (ggml_backend_dev_t);
    const char *               (*get_description)     (ggml_backend_dev_t);
    void                       (*get_memory)          (ggml_backend_dev_t, size_t * free, size_t * total);
    enum ggml_backend_dev_type (*get_type)            (ggml_backend_dev_t);
    void                       (*get_props)           (ggml_backend_dev_t, struct ggml_backend_dev_props *);
    ggml_backend_t             (*init_backend)        (ggml_backend_dev_t, const char * params);
    ggml_backend_buffer_type_t (*get_buffer_type)     (ggml_backend_dev_t);
    ggml_backend_buffer_type_t (*get_host_buffer_type)(ggml_backend_dev_t);
    ggml_backend_buffer_t      (*buffer_from_host_ptr)(ggml_backend_dev_t, void * ptr, size_t size, size_t max_tensor_size);
    bool                       (*supports_op)         (ggml_backend_dev_t, const struct ggml_tensor * op);
    bool                       (*supports_buft)       (ggml_backend_dev_t, ggml_backend_buffer_type_t);
    bool                       (*offload_op)          (ggml_backend_dev_t, const struct ggml_tensor * op);
    ggml_backend_event_t       (*event_new)           (ggml_backend_dev_t);
    void                       (*event_free)          (ggml_backend_dev_t, ggml_backend_event_t);
    void                       (*event_synchronize)   (ggml_backend_dev_t, ggml_backend_event_t);
};
struct ggml_backend_device {            // :184-188
    struct ggml_backend_device_i iface;
    ggml_backend_reg_t           reg;
    void *                       context;
};

struct ggml_backend_reg_i {             // :194-204
    const char *       (*get_name)        (ggml_backend_reg_t);
    size_t             (*get_device_count)(ggml_backend_reg_t);
    ggml_backend_dev_t (*get_device)      (ggml_backend_reg_t, size_t index);
    void *             (*get_proc_address)(ggml_backend_reg_t, const char * name);
};
struct ggml_backend_reg {               // :206-210
    int                       api_version;
    struct ggml_backend_reg_i iface;
    void *                    context;
};

// ---------------------------------------------------------------- quantised block formats (ggml-common.h)
typedef uint16_t ggml_half;
#define QK_K   256
#define QK8_0  32
#define K_SCALE_SIZE 12
#pragma pack(push, 1)
typedef struct { ggml_half d; int8_t qs[QK8_0]; } block_q8_0;                                   // 34 B  (:219-224)
typedef struct { ggml_half d; ggml_half dmin; uint8_t scales[K_SCALE_SIZE]; uint8_t qs[QK_K/2]; } block_q4_K;   // 144 B (:295-305)
typedef struct { uint8_t ql[QK_K/2]; uint8_t qh[QK_K/4]; int8_t scales[QK_K/16]; ggml_half d; } block_q6_K;     // 210 B (:330-335)
#pragma pack(pop)
typedef struct { float d; int8_t qs[QK_K]; int16_t bsums[QK_K/16]; } block_q8_K;                // 292 B (:339-343)

#ifdef __cplusplus
}
static_assert(sizeof(ggml_tensor) == 336, "ggml_tensor ABI");
static_assert(sizeof(ggml_cgraph) == 88, "ggml_cgraph ABI");
static_assert(sizeof(ggml_backend_i) == 112 && sizeof(ggml_backend_buffer_i) == 72 && sizeof(ggml_backend_buffer_type_i) == 48, "vtable ABI");
static_assert(sizeof(ggml_backend_device_i) == 120 && sizeof(ggml_backend_reg_i) == 32, "vtable ABI");
static_assert(sizeof(ggml_backend) == 136 && sizeof(ggml_backend_buffer) == 104 && sizeof(ggml_backend_buffer_type) == 64, "object ABI");
static_assert(sizeof(ggml_backend_device) == 136 && sizeof(ggml_backend_reg) == 48, "object ABI");
static_assert(sizeof(block_q8_0) == 34 && sizeof(block_q4_K) == 144 && sizeof(block_q6_K) == 210 && sizeof(block_q8_K) == 292, "block ABI");
#endif
