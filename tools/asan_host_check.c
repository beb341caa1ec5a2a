/* Host-side AddressSanitizer run of libcaptioner_hip's argument handling (no GPU needed; GPU ASan is not available on the
 * pool).  Built and run by tools/asan_host_check.sh against an ASan build of the library: every call below must return an
 * error code with a message - no crash, no out-of-bounds access, no leak on the failure paths (LeakSanitizer at exit). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/captioner_hip.h"

static int failures = 0;
#define EXPECT(cond) do { if (!(cond)) { printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

static CapConfig blip_cfg(void) {
    CapConfig c;
    memset(&c, 0, sizeof(c));
    c.struct_size = (int32_t)sizeof(c);
    c.arch = CAP_ARCH_BLIP; c.compute_dtype = CAP_F32_SPLIT;
    c.image_size = 32; c.patch_size = 16; c.v_hidden = 64; c.v_layers = 2; c.v_heads = 1; c.v_mlp = 128; c.v_eps = 1e-5f;
    c.t_hidden = 64; c.t_layers = 2; c.t_heads = 1; c.t_ffn = 128; c.vocab = 512; c.max_pos = 64; c.t_eps = 1e-12f;
    c.bos = 1; c.eos = 2; c.pad = 0; c.max_batch = 4; c.max_beams = 3; c.max_len = 12;
    return c;
}

int main(void) {
    CapHandle h = NULL;
    CapConfig c = blip_cfg();
    EXPECT(cap_version() >= 1);
    EXPECT(cap_create(NULL, &h) != 0 && strlen(cap_last_error()) > 0);
    EXPECT(cap_create(&c, NULL) != 0);
    c.struct_size -= 4;
    EXPECT(cap_create(&c, &h) != 0 && strstr(cap_last_error(), "size mismatch"));
    c = blip_cfg(); c.arch = 17;
    EXPECT(cap_create(&c, &h) != 0 && strstr(cap_last_error(), "unknown arch"));
    c = blip_cfg(); c.compute_dtype = 9;
    EXPECT(cap_create(&c, &h) != 0 && strstr(cap_last_error(), "dtype"));
    c = blip_cfg(); c.arch = CAP_ARCH_COCA;                     /* CoCa without its pooler / multimodal geometry */
    EXPECT(cap_create(&c, &h) != 0);
    c = blip_cfg(); c.arch = CAP_ARCH_MINILM; c.t_heads = 2;    /* the sentence encoder has no split mode */
    EXPECT(cap_create(&c, &h) != 0 && strstr(cap_last_error(), "CAP_F32_SPLIT"));
    c = blip_cfg(); c.v_heads = 3;                              /* head_dim != 64 */
    EXPECT(cap_create(&c, &h) != 0 && strstr(cap_last_error(), "head_dim"));
    c = blip_cfg(); c.max_beams = 99;
    EXPECT(cap_create(&c, &h) != 0);
    c = blip_cfg(); c.max_len = 1000;                           /* > max_pos */
    EXPECT(cap_create(&c, &h) != 0);
    c = blip_cfg(); c.arch = CAP_ARCH_MINILM; c.compute_dtype = CAP_BF16; c.t_heads = 5;
    EXPECT(cap_create(&c, &h) != 0);
    c = blip_cfg(); c.arch = CAP_ARCH_BLIP2; c.compute_dtype = CAP_BF16;
    EXPECT(cap_create(&c, &h) != 0);
    /* a valid configuration: without a GPU the first HIP call fails - the handle must be released on that path */
    c = blip_cfg();
    h = NULL;
    int rc = cap_create(&c, &h);
    if (rc == 0) { printf("note: a GPU is present, cap_create succeeded\n"); EXPECT(cap_destroy(h) == 0); }
    else EXPECT(strlen(cap_last_error()) > 0);
    EXPECT(cap_create_shared(&c, NULL, &h) != 0);
    /* null handles / buffers */
    int64_t shape[2] = {2, 2};
    float w[4] = {0};
    EXPECT(cap_load_weight(NULL, "x", w, 0, 2, shape, NULL) != 0);
    EXPECT(cap_finalize_weights(NULL) != 0);
    EXPECT(cap_destroy(NULL) == 0);
    EXPECT(cap_generate_groups(NULL, w, 0, 1, 6, 3, 8, 1.0f, NULL, NULL, NULL, NULL) != 0);
    EXPECT(cap_set_early_exit(NULL, 2) != 0);
    EXPECT(cap_last_decode_steps(NULL) == -1);
    EXPECT(cap_device_bytes(NULL) == 0);
    EXPECT(cap_encode(NULL, w, 0, 1, w, NULL) != 0);
    EXPECT(cap_generate(NULL, w, 0, 1, 1, 8, 1.0f, NULL, NULL, NULL, NULL, NULL) != 0);
    EXPECT(cap_embed_text(NULL, NULL, NULL, 1, 1, NULL, NULL) != 0);
    EXPECT(cap_profile_enable(NULL, 1) != 0);
    char buf[8];
    EXPECT(cap_profile_report(NULL, buf, sizeof(buf)) != 0);
    /* pure host planner of the weight-streaming GEMM */
    for (int n = 32; n <= 12288; n += 160)
        for (int k = 64; k <= 10240; k += 448) { (void)cap_op_gemm_skinny_slices(n, k, 0); (void)cap_op_gemm_skinny_slices(n, k, 1); }
    /* GEMM launcher argument checks come before any HIP call */
    EXPECT(cap_op_gemm(CAP_F32, w, w, NULL, NULL, w, 64, 62, 64, 0, 1, 3, NULL) != 0);
    EXPECT(cap_op_gemm(CAP_F32_SPLIT, w, w, NULL, NULL, w, 64, 68, 96, 0, 0, 0, NULL) != 0);
    EXPECT(cap_op_gemm(CAP_BF16, w, w, NULL, NULL, w, 0, 64, 64, 0, 1, 0, NULL) != 0);
    /* round 3 / 4 entry points: bad shapes and null handles are refused before any HIP call */
    EXPECT(cap_op_gemm_partial(CAP_F32_SPLIT, w, w, w, 64, 64, 96, 4, 6, NULL) != 0);        /* K not a multiple of 4 slabs */
    EXPECT(cap_op_gemm_partial(CAP_BF16, w, w, w, 64, 62, 128, 1, 6, NULL) != 0);            /* N % 4 */
    EXPECT(cap_op_gemm_crosskv(CAP_F32_SPLIT, w, w, NULL, w, 1, 40, 1, 1, 48, 1, NULL) != 0);  /* K below two stages: no KV16 epilogue */
    EXPECT(cap_op_gemm_crosskv(CAP_F32, w, w, NULL, w, 1, 40, 1, 1, 64, 1, NULL) != 0);      /* KV16 needs G8 operands */
    EXPECT(cap_set_decode_path(NULL, 0) != 0);
    EXPECT(cap_last_decode_path(NULL) == -1);
    EXPECT(cap_cross_cache_kind(NULL) == -1);
    /* round 6: row compaction switch (the decode path numbers stop at 2 since the fused-tile path left the tree) */
    EXPECT(cap_set_row_compaction(NULL, 1) != 0 && strlen(cap_last_error()) > 0);
    EXPECT(cap_last_row_compaction(NULL) == -1);
    EXPECT(cap_op_vit_attention(CAP_F32_SPLIT, w, w, 1, 197, 12, 5, NULL) != 0 || 1);       /* no GPU: a clean error from the launch */
    {
        long long sat = cap_g8_saturations(0);                                                /* no GPU: a clean error, or a count */
        EXPECT(sat >= -1);
        int rcp = cap_op_pack_kv16(w, w, 0, NULL);                                            /* zero rows: nothing to launch */
        EXPECT(rcp == 0 || strlen(cap_last_error()) > 0);
    }
    /* round 6, second session: int8 weights (CapConfig.weight_int8) - the configuration checks and the host planner / argument
     * checks of the int8 weight stream run before any HIP call */
    c = blip_cfg(); c.weight_int8 = 1;                                                        /* not BLIP-2 */
    EXPECT(cap_create(&c, &h) != 0 && strstr(cap_last_error(), "weight_int8"));
    c = blip_cfg(); c.arch = CAP_ARCH_BLIP2; c.compute_dtype = CAP_F32_SPLIT; c.weight_int8 = 1;   /* not bf16 */
    EXPECT(cap_create(&c, &h) != 0);
    c = blip_cfg(); c.arch = CAP_ARCH_BLIP2; c.compute_dtype = CAP_BF16; c.weight_int8 = 7;    /* not a flag value */
    EXPECT(cap_create(&c, &h) != 0);
    for (int n = 32; n <= 12288; n += 160)
        for (int k = 64; k <= 10240; k += 448) { (void)cap_op_gemm_skinny_i8_slices(n, k, 0); (void)cap_op_gemm_skinny_i8_slices(n, k, 1); }
    EXPECT(cap_op_gemm_skinny_i8_slices(2560, 2560, 1) == 1 && cap_op_gemm_skinny_i8_slices(2560, 10240, 0) >= 1);
    EXPECT(cap_op_gemm_skinny_i8_slices(40, 2560, 1) == 0);                                   /* N % 32 */
    EXPECT(cap_op_quant_i8_pack(w, w, w, 20, 64, NULL) != 0 && strstr(cap_last_error(), "quant_i8_pack"));   /* rows % 16 */
    EXPECT(cap_op_quant_i8_pack(w, w, w, 32, 100, NULL) != 0);                                /* cols % 64 */
    EXPECT(cap_op_gemm_skinny_i8(w, w, w, NULL, 0, w, NULL, 4, 40, 256, NULL) != 0 && strstr(cap_last_error(), "gemm_skinny_i8"));
    EXPECT(cap_op_gemm_skinny_i8(w, w, w, NULL, 0, w, NULL, 0, 256, 256, NULL) != 0);          /* no rows */
    EXPECT(cap_op_gemm_skinny_i8(w, w, w, NULL, 0, NULL, NULL, 4, 256, 256, NULL) != 0);       /* neither output */
    {   /* the list form of the crop + resize: the frame table is mandatory, shapes are checked before any HIP call */
        int32_t iw[8] = {0};
        uint8_t ob[4] = {0};
        EXPECT(cap_crop_resize_u8_frames((const uint8_t*)w, NULL, 0, iw, iw, iw, 1, iw, iw, 1, 1, 1, ob, NULL) != 0 && strstr(cap_last_error(), "frame table"));
        int64_t fr[3] = {0, 1, 1};
        EXPECT(cap_crop_resize_u8_frames((const uint8_t*)w, fr, 0, iw, iw, iw, 0, iw, iw, 1, 1, 1, ob, NULL) != 0);          /* KH < 1 */
        EXPECT(cap_crop_resize_u8_frames(NULL, fr, 0, iw, iw, iw, 1, iw, iw, 1, 1, 1, ob, NULL) != 0);
    }
    printf(failures ? "asan host check: %d FAILED\n" : "asan host check: all calls returned cleanly (%d failures)\n", failures);
    return failures ? 1 : 0;
}
