/* zkr_napi.c -- thin N-API addon binding the C ABI of include/zkr.h for Node.js.
 *
 * It is the FFI a maintainer of kendricktan/simple-zk-rollups adds so that
 *   import { buildBn128 } from "websnark"            (operator/src/snarks/common.ts:5, scripts/index.js:19)
 * can become  require("simple-zk-rollups_amd")  with the same surface (see ../index.js).
 * Plain C against <node_api.h>; libzkr_hip.so is dlopen'ed at run time, so the addon itself loads on a
 * machine without ROCm and reports the problem as a rejected promise / thrown Error.
 */
#include <dlfcn.h>
#include <node_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct zkr_key zkr_key;
static struct {
  void *handle;
  const char *(*last_error)(void);
  const char *(*version)(void);
  int (*device_count)(void);
  int (*key_load_websnark)(const void *, size_t, int, zkr_key **);
  void (*key_free)(zkr_key *);
  int (*key_info)(const zkr_key *, uint64_t *);
  int (*prove)(zkr_key *, const void *, size_t, const uint8_t *, const uint8_t *, uint8_t *, void *);
  int (*prove_batch)(zkr_key *, const void *const *, size_t, size_t, const uint8_t *, const uint8_t *, uint8_t *);
  int (*prove_batch_multi)(zkr_key *const *, size_t, const void *const *, size_t, size_t, const uint8_t *, const uint8_t *, uint8_t *);
  int (*key_replicate)(const zkr_key *, int, int, zkr_key **);
  int (*key_shard)(const zkr_key *, unsigned, unsigned, int, zkr_key **);
  int (*prove_sharded)(zkr_key *const *, size_t, const void *, size_t, const uint8_t *, const uint8_t *, uint8_t *);
  int (*key_device)(const zkr_key *);
  int (*verify)(const void *, size_t, const uint8_t *, const void *, size_t, int *);
  int (*verify_batch)(const void *, size_t, const uint8_t *, const void *, size_t, size_t, int *);
  int (*setup_r1cs)(const void *, size_t, const uint8_t *, int, zkr_key **, void **, size_t *);
  int (*key_save)(const zkr_key *, const char *);
  int (*key_load_file)(const char *, int, zkr_key **);
  void (*free_)(void *);
  int (*multihash)(const uint8_t *, size_t, uint8_t *);
  int (*pubkey)(const uint8_t *, uint8_t *);
  int (*eddsa_sign)(const uint8_t *, const uint8_t *, size_t, uint8_t *);
  int (*eddsa_verify)(const uint8_t *, size_t, const uint8_t *, const uint8_t *, int *);
  int (*format_privkey)(const uint8_t *, uint8_t *);
  int (*multihash_batch)(const void *, size_t, unsigned, void *, int);
  int (*tree_build)(const void *, unsigned, void *, int);
  int (*withdraw_r1cs)(void **, size_t *);
  int (*withdraw_witness)(const uint8_t *, const uint8_t *, void **, size_t *);
  int (*rollup_info)(uint32_t, uint32_t, uint32_t *, uint32_t *, uint32_t *);
  int (*rollup_r1cs)(uint32_t, uint32_t, void **, size_t *);
  int (*rollup_witness)(uint32_t, uint32_t, const uint8_t *, size_t, void **, size_t *);
  int (*sharded_last_form)(int *, char *, size_t);
  int (*key_replication)(const zkr_key *, int *, int *);
} Z;
/* which form the last sharded proof took (zkr_prove_sharded_last_form, read on the worker thread that ran it, published on the JS
 * thread when its promise settles) */
static int g_sharded_form = 0;
static char g_sharded_reason[256] = "";

#define NAPI_OK(call)                                                     \
  do {                                                                    \
    if ((call) != napi_ok) {                                              \
      napi_throw_error(env, NULL, "zkr_napi: N-API call failed: " #call); \
      return NULL;                                                        \
    }                                                                     \
  } while (0)

static napi_value throw_msg(napi_env env, const char *msg) {
  napi_throw_error(env, "ZKR", msg);
  return NULL;
}

/* load(libPath) -> number of HIP devices */
static napi_value js_load(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  char path[4096];
  size_t n = 0;
  NAPI_OK(napi_get_value_string_utf8(env, argv[0], path, sizeof(path), &n));
  if (!Z.handle) {
    void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) {
      char msg[4600];
      strcpy(msg, "cannot load libzkr_hip.so (there is no CPU fallback): ");
      strncat(msg, dlerror(), 4000);
      return throw_msg(env, msg);
    }
#define SYM(field, name)                                                        \
  *(void **)(&Z.field) = dlsym(h, name);                                        \
  if (!Z.field) return throw_msg(env, "libzkr_hip.so lacks symbol " name);
    SYM(last_error, "zkr_last_error") SYM(version, "zkr_version") SYM(device_count, "zkr_device_count")
    SYM(key_load_websnark, "zkr_key_load_websnark") SYM(key_free, "zkr_key_free") SYM(key_info, "zkr_key_info") SYM(prove, "zkr_prove") SYM(prove_batch, "zkr_prove_batch") SYM(prove_batch_multi, "zkr_prove_batch_multi") SYM(key_replicate, "zkr_key_replicate") SYM(key_shard, "zkr_key_shard") SYM(prove_sharded, "zkr_prove_sharded") SYM(key_device, "zkr_key_device") SYM(verify, "zkr_verify") SYM(verify_batch, "zkr_verify_batch")
    SYM(setup_r1cs, "zkr_setup_r1cs") SYM(key_save, "zkr_key_save") SYM(key_load_file, "zkr_key_load_file") SYM(free_, "zkr_free")
    SYM(multihash, "zkr_mimcsponge_multihash") SYM(pubkey, "zkr_babyjub_pubkey") SYM(eddsa_sign, "zkr_eddsa_sign") SYM(eddsa_verify, "zkr_eddsa_verify")
    SYM(multihash_batch, "zkr_mimcsponge_multihash_batch") SYM(tree_build, "zkr_balance_tree_build")
    SYM(format_privkey, "zkr_babyjub_format_privkey") SYM(withdraw_r1cs, "zkr_withdraw_r1cs") SYM(withdraw_witness, "zkr_withdraw_witness")
    SYM(rollup_info, "zkr_rollup_info") SYM(rollup_r1cs, "zkr_rollup_r1cs") SYM(rollup_witness, "zkr_rollup_witness")
    SYM(sharded_last_form, "zkr_prove_sharded_last_form") SYM(key_replication, "zkr_key_replication")
    Z.handle = h;
  }
  napi_value out;
  NAPI_OK(napi_create_int32(env, Z.device_count(), &out));
  return out;
}

static napi_value js_version(napi_env env, napi_callback_info info) {
  (void)info;
  if (!Z.handle) return throw_msg(env, "libzkr_hip.so not loaded");
  napi_value out;
  NAPI_OK(napi_create_string_utf8(env, Z.version(), NAPI_AUTO_LENGTH, &out));
  return out;
}

static int get_bytes(napi_env env, napi_value v, const uint8_t **data, size_t *len) {
  bool is;
  if (napi_is_arraybuffer(env, v, &is) == napi_ok && is) return napi_get_arraybuffer_info(env, v, (void **)data, len) == napi_ok;
  if (napi_is_buffer(env, v, &is) == napi_ok && is) return napi_get_buffer_info(env, v, (void **)data, len) == napi_ok;
  if (napi_is_typedarray(env, v, &is) == napi_ok && is) {
    napi_typedarray_type t;
    napi_value ab;
    size_t off, count;
    if (napi_get_typedarray_info(env, v, &t, &count, (void **)data, &ab, &off) != napi_ok) return 0;
    *len = count * (t == napi_uint8_array || t == napi_int8_array || t == napi_uint8_clamped_array ? 1 : t == napi_uint32_array || t == napi_int32_array || t == napi_float32_array ? 4 : 8);
    return 1;
  }
  return 0;
}

static void key_finalize(napi_env env, void *data, void *hint) {
  (void)env; (void)hint;
  if (data && Z.key_free) Z.key_free((zkr_key *)data);
}

/* keyLoad(provingKeyBin, device) -> external handle (synchronous: parse + upload once per key) */
static napi_value js_key_load(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (!Z.handle) return throw_msg(env, "libzkr_hip.so not loaded");
  const uint8_t *pk; size_t len; int32_t dev = 0;
  if (!get_bytes(env, argv[0], &pk, &len)) return throw_msg(env, "provingKeyBin must be an ArrayBuffer / Buffer / TypedArray");
  if (argc > 1) napi_get_value_int32(env, argv[1], &dev);
  zkr_key *key = NULL;
  int rc = Z.key_load_websnark(pk, len, dev, &key);
  if (rc) return throw_msg(env, Z.last_error());
  napi_value ext;
  NAPI_OK(napi_create_external(env, key, key_finalize, NULL, &ext));
  return ext;
}

/* keyReplicate(key, device, mode) -> external handle of an independent replica on `device` (zkr_key_replicate: device-to-device
 * copy of the key arena inside this process; mode 0 auto, 1 full arena, 2 compact arena + rebuild).  Synchronous, once per
 * (key, device). */
static napi_value js_key_replicate(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value argv[3];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (!Z.handle) return throw_msg(env, "libzkr_hip.so not loaded");
  zkr_key *src = NULL, *key = NULL;
  int32_t dev = 0, mode = 0;
  if (argc < 2 || napi_get_value_external(env, argv[0], (void **)&src) != napi_ok || napi_get_value_int32(env, argv[1], &dev) != napi_ok)
    return throw_msg(env, "keyReplicate(key, device, mode)");
  if (argc > 2) napi_get_value_int32(env, argv[2], &mode);
  if (Z.key_replicate(src, dev, mode, &key)) return throw_msg(env, Z.last_error());
  napi_value ext;
  NAPI_OK(napi_create_external(env, key, key_finalize, NULL, &ext));
  return ext;
}
/* keyShard(key, part, parts, device) -> external handle of shard `part` of `parts` of the whole key on `device` (zkr_key_shard: one
 * contiguous range of every MSM of a proof, all window levels copied device to device, plus the whole QAP).  Synchronous. */
static napi_value js_key_shard(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (!Z.handle) return throw_msg(env, "libzkr_hip.so not loaded");
  zkr_key *src = NULL, *key = NULL;
  uint32_t part = 0, parts = 0;
  int32_t dev = 0;
  if (argc < 4 || napi_get_value_external(env, argv[0], (void **)&src) != napi_ok || napi_get_value_uint32(env, argv[1], &part) != napi_ok ||
      napi_get_value_uint32(env, argv[2], &parts) != napi_ok || napi_get_value_int32(env, argv[3], &dev) != napi_ok)
    return throw_msg(env, "keyShard(key, part, parts, device)");
  if (Z.key_shard(src, part, parts, dev, &key)) return throw_msg(env, Z.last_error());
  napi_value ext;
  NAPI_OK(napi_create_external(env, key, key_finalize, NULL, &ext));
  return ext;
}
/* keyDevice(key) -> HIP ordinal of the key's device */
static napi_value js_key_device(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  zkr_key *key = NULL;
  if (argc < 1 || napi_get_value_external(env, argv[0], (void **)&key) != napi_ok) return throw_msg(env, "keyDevice(key)");
  napi_value out;
  NAPI_OK(napi_create_int32(env, Z.key_device(key), &out));
  return out;
}

/* setupR1cs(r1csBin, toxic160|null, device) -> [key handle, vkBin Buffer]: Groth16 setup on the GPU (zkr_setup_r1cs;
 * `snarkjs setup --protocol groth`, prover/package.json:34,37).  Synchronous: done once per circuit. */
static napi_value js_setup_r1cs(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value argv[3];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (!Z.handle) return throw_msg(env, "libzkr_hip.so not loaded");
  const uint8_t *r1cs, *tox = NULL;
  size_t len, tox_len = 0;
  int32_t dev = 0;
  if (argc < 1 || !get_bytes(env, argv[0], &r1cs, &len)) return throw_msg(env, "r1csBin must be an ArrayBuffer / Buffer / TypedArray");
  napi_valuetype t = napi_null;
  if (argc > 1) napi_typeof(env, argv[1], &t);
  if (t != napi_null && t != napi_undefined) {
    if (!get_bytes(env, argv[1], &tox, &tox_len) || tox_len != 160) return throw_msg(env, "toxic waste must be 5 x 32 bytes or null");
  }
  if (argc > 2) napi_get_value_int32(env, argv[2], &dev);
  zkr_key *key = NULL;
  void *vk = NULL;
  size_t vk_len = 0;
  if (Z.setup_r1cs(r1cs, len, tox, dev, &key, &vk, &vk_len)) return throw_msg(env, Z.last_error());
  napi_value ext, vkb, arr;
  void *copy;
  NAPI_OK(napi_create_external(env, key, key_finalize, NULL, &ext));
  NAPI_OK(napi_create_buffer_copy(env, vk_len, vk, &copy, &vkb));
  Z.free_(vk);
  NAPI_OK(napi_create_array_with_length(env, 2, &arr));
  NAPI_OK(napi_set_element(env, arr, 0, ext));
  NAPI_OK(napi_set_element(env, arr, 1, vkb));
  return arr;
}

/* keySave(key, path) / keyLoadFile(path, device): the packed device-layout key file (zkr_key_save / zkr_key_load_file) */
static napi_value js_key_save(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  zkr_key *key;
  char path[4096];
  size_t n = 0;
  NAPI_OK(napi_get_value_external(env, argv[0], (void **)&key));
  NAPI_OK(napi_get_value_string_utf8(env, argv[1], path, sizeof(path), &n));
  if (Z.key_save(key, path)) return throw_msg(env, Z.last_error());
  return NULL;
}
static napi_value js_key_load_file(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (!Z.handle) return throw_msg(env, "libzkr_hip.so not loaded");
  char path[4096];
  size_t n = 0;
  int32_t dev = 0;
  NAPI_OK(napi_get_value_string_utf8(env, argv[0], path, sizeof(path), &n));
  if (argc > 1) napi_get_value_int32(env, argv[1], &dev);
  zkr_key *key = NULL;
  if (Z.key_load_file(path, dev, &key)) return throw_msg(env, Z.last_error());
  napi_value ext;
  NAPI_OK(napi_create_external(env, key, key_finalize, NULL, &ext));
  return ext;
}

static napi_value js_key_info(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  zkr_key *key;
  NAPI_OK(napi_get_value_external(env, argv[0], (void **)&key));
  uint64_t v[10];
  if (Z.key_info(key, v)) return throw_msg(env, Z.last_error());
  napi_value arr;
  NAPI_OK(napi_create_array_with_length(env, 10, &arr));
  for (uint32_t i = 0; i < 10; i++) {
    napi_value x;
    NAPI_OK(napi_create_double(env, (double)v[i], &x));
    NAPI_OK(napi_set_element(env, arr, i, x));
  }
  return arr;
}

typedef struct {
  napi_async_work work;
  napi_deferred deferred;
  napi_ref key_ref, wit_ref;
  zkr_key *key;
  zkr_key **shards;   /* proveSharded: the shards of one key, part i at position i (zkr_prove_sharded); NULL for prove */
  size_t n_shards;
  const uint8_t *wit;
  size_t wit_len;
  int have_rs;
  uint8_t r[32], s[32], proof[256];
  int rc;
  char err[512];
  int form;            /* proveSharded: split / replicated calcH and why */
  char reason[256];
} prove_job;

static void prove_execute(napi_env env, void *data) {  /* libuv worker thread: the event loop is not blocked */
  (void)env;
  prove_job *j = (prove_job *)data;
  j->rc = j->shards ? Z.prove_sharded(j->shards, j->n_shards, j->wit, j->wit_len, j->have_rs ? j->r : NULL, j->have_rs ? j->s : NULL, j->proof)
                    : Z.prove(j->key, j->wit, j->wit_len, j->have_rs ? j->r : NULL, j->have_rs ? j->s : NULL, j->proof, NULL);
  if (j->rc) { strncpy(j->err, Z.last_error(), sizeof(j->err) - 1); j->err[sizeof(j->err) - 1] = 0; }
  if (j->shards) Z.sharded_last_form(&j->form, j->reason, sizeof(j->reason));  /* thread-local in the library: this thread ran the proof */
}

static void prove_complete(napi_env env, napi_status status, void *data) {
  prove_job *j = (prove_job *)data;
  napi_value v;
  if (j->shards && status == napi_ok) { g_sharded_form = j->form; memcpy(g_sharded_reason, j->reason, sizeof(g_sharded_reason)); }
  if (status == napi_ok && j->rc == 0) {
    void *out;
    napi_create_buffer_copy(env, 256, j->proof, &out, &v);
    napi_resolve_deferred(env, j->deferred, v);
  } else {
    napi_value msg;
    napi_create_string_utf8(env, j->rc ? j->err : "proof job cancelled", NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, NULL, msg, &v);
    napi_reject_deferred(env, j->deferred, v);
  }
  napi_delete_reference(env, j->key_ref);
  napi_delete_reference(env, j->wit_ref);
  napi_delete_async_work(env, j->work);
  free(j->shards);
  free(j);
}

/* prove(key, witnessBin, r32|null, s32|null) -> Promise<Buffer(256)>; the witness buffer is pinned by a
 * reference until the promise settles (the reference's caller may reuse it only afterwards).
 * proveSharded([shard0, shard1, ...], witnessBin, r32|null, s32|null): ONE proof over the shards of a key (zkr_prove_sharded: one
 * host thread per shard, 640 bytes of partial sums each, assembled on the host); the array of handles is referenced until the
 * promise settles. */
static napi_value prove_common(napi_env env, napi_callback_info info, int sharded) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (!Z.handle) return throw_msg(env, "libzkr_hip.so not loaded");
  if (argc < 2) return throw_msg(env, "prove(key, witnessBin, r, s) / proveSharded(shards, witnessBin, r, s)");
  prove_job *j = (prove_job *)calloc(1, sizeof(prove_job));
  if (!j) return throw_msg(env, "prove: out of memory");
  if (sharded) {
    uint32_t ns = 0;
    if (napi_get_array_length(env, argv[0], &ns) != napi_ok || ns == 0 || !(j->shards = (zkr_key **)calloc(ns, sizeof(zkr_key *)))) { free(j); return throw_msg(env, "proveSharded: a non-empty array of shard handles expected"); }
    j->n_shards = ns;
    for (uint32_t i = 0; i < ns; i++) {
      napi_value kv;
      if (napi_get_element(env, argv[0], i, &kv) != napi_ok || napi_get_value_external(env, kv, (void **)&j->shards[i]) != napi_ok) { free(j->shards); free(j); return throw_msg(env, "proveSharded: every entry must be a key handle"); }
    }
  } else if (napi_get_value_external(env, argv[0], (void **)&j->key) != napi_ok) { free(j); return throw_msg(env, "prove: a key handle expected"); }
  if (!get_bytes(env, argv[1], &j->wit, &j->wit_len)) { free(j->shards); free(j); return throw_msg(env, "witnessBin must be an ArrayBuffer / Buffer / TypedArray"); }
  const uint8_t *p; size_t n;
  if (argc >= 4 && get_bytes(env, argv[2], &p, &n) && n == 32) {
    memcpy(j->r, p, 32);
    if (!get_bytes(env, argv[3], &p, &n) || n != 32) { free(j->shards); free(j); return throw_msg(env, "r and s must both be 32-byte buffers"); }
    memcpy(j->s, p, 32);
    j->have_rs = 1;
  }
  /* nothing below may leave the job, its references or its work item behind when an N-API call fails (nothing was queued then) */
  napi_value promise, name;
  if (napi_create_promise(env, &j->deferred, &promise) != napi_ok || napi_create_reference(env, argv[0], 1, &j->key_ref) != napi_ok ||
      napi_create_reference(env, argv[1], 1, &j->wit_ref) != napi_ok || napi_create_string_utf8(env, "zkr_prove", NAPI_AUTO_LENGTH, &name) != napi_ok ||
      napi_create_async_work(env, NULL, name, prove_execute, prove_complete, j, &j->work) != napi_ok || napi_queue_async_work(env, j->work) != napi_ok) {
    if (j->deferred) {  /* the promise exists but nothing will ever settle it: reject it before the job goes (it is not returned) */
      napi_value msg, err;
      if (napi_create_string_utf8(env, "prove: could not queue the proof job", NAPI_AUTO_LENGTH, &msg) == napi_ok && napi_create_error(env, NULL, msg, &err) == napi_ok)
        napi_reject_deferred(env, j->deferred, err);
    }
    if (j->key_ref) napi_delete_reference(env, j->key_ref);
    if (j->wit_ref) napi_delete_reference(env, j->wit_ref);
    if (j->work) napi_delete_async_work(env, j->work);
    free(j->shards);
    free(j);
    return throw_msg(env, "prove: could not queue the proof job");
  }
  return promise;
}
/* shardedLastForm() -> {form: "none" | "split" | "replicated", reason}: of the last proveSharded whose promise has settled */
static napi_value js_sharded_last_form(napi_env env, napi_callback_info info) {
  (void)info;
  static const char *names[] = {"none", "split", "replicated"};
  napi_value out, f, r;
  NAPI_OK(napi_create_object(env, &out));
  NAPI_OK(napi_create_string_utf8(env, names[g_sharded_form >= 0 && g_sharded_form <= 2 ? g_sharded_form : 0], NAPI_AUTO_LENGTH, &f));
  NAPI_OK(napi_create_string_utf8(env, g_sharded_reason, NAPI_AUTO_LENGTH, &r));
  NAPI_OK(napi_set_named_property(env, out, "form", f));
  NAPI_OK(napi_set_named_property(env, out, "reason", r));
  return out;
}
/* keyReplication(key) -> {mode: "none" | "full" | "base", peerDirect}: how the key came to its device (zkr_key_replication) */
static napi_value js_key_replication(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (!Z.handle) return throw_msg(env, "libzkr_hip.so not loaded");
  zkr_key *key = NULL;
  if (argc < 1 || napi_get_value_external(env, argv[0], (void **)&key) != napi_ok) return throw_msg(env, "keyReplication(key)");
  int mode = 0, direct = 0;
  if (Z.key_replication(key, &mode, &direct)) return throw_msg(env, Z.last_error());
  static const char *names[] = {"none", "full", "base"};
  napi_value out, m, d;
  NAPI_OK(napi_create_object(env, &out));
  NAPI_OK(napi_create_string_utf8(env, names[mode >= 0 && mode <= 2 ? mode : 0], NAPI_AUTO_LENGTH, &m));
  NAPI_OK(napi_get_boolean(env, direct != 0, &d));
  NAPI_OK(napi_set_named_property(env, out, "mode", m));
  NAPI_OK(napi_set_named_property(env, out, "peerDirect", d));
  return out;
}
static napi_value js_prove(napi_env env, napi_callback_info info) { return prove_common(env, info, 0); }
static napi_value js_prove_sharded(napi_env env, napi_callback_info info) { return prove_common(env, info, 1); }

/* proveBatch(key, [witnessBin...], rs|null, ss|null) -> Promise<Buffer(256 * count)>: zkr_prove_batch on a libuv worker --
 * uploads and proofs pipelined over the key's two workspaces, proofs of small circuits (the reference's tx circuit) fused
 * into shared launches.  rs / ss: count x 32 bytes of blinding scalars or both null (drawn per proof). */
typedef struct {
  napi_async_work work;
  napi_deferred deferred;
  napi_ref key_ref;    /* the key handle, or the JS array of replica handles (proveBatchMulti): alive until the job is done */
  napi_ref *wit_refs;  /* one strong reference per witness buffer: the worker reads their memory, whatever the caller does to the array meanwhile */
  zkr_key *key;
  zkr_key **keys;      /* proveBatchMulti: replicas of one key, one per device (zkr_prove_batch_multi); NULL for proveBatch */
  size_t n_keys;
  const void **wits;
  size_t count, wit_len, n_refs;
  uint8_t *rs, *ss, *proofs;
  int rc;
  char err[512];
} batch_job;

/* everything a job holds, on the JS thread (references need the env); safe on a partly built job */
static void batch_free(napi_env env, batch_job *j) {
  if (!j) return;
  if (j->key_ref) napi_delete_reference(env, j->key_ref);
  for (size_t i = 0; i < j->n_refs; i++) napi_delete_reference(env, j->wit_refs[i]);
  if (j->work) napi_delete_async_work(env, j->work);
  free(j->wit_refs); free(j->wits); free(j->keys); free(j->rs); free(j->ss); free(j->proofs); free(j);
}
static void batch_execute(napi_env env, void *data) {
  (void)env;
  batch_job *j = (batch_job *)data;
  j->rc = j->keys ? Z.prove_batch_multi(j->keys, j->n_keys, j->wits, j->wit_len, j->count, j->rs, j->ss, j->proofs)
                  : Z.prove_batch(j->key, j->wits, j->wit_len, j->count, j->rs, j->ss, j->proofs);
  if (j->rc) { strncpy(j->err, Z.last_error(), sizeof(j->err) - 1); j->err[sizeof(j->err) - 1] = 0; }
}
static void batch_complete(napi_env env, napi_status status, void *data) {
  batch_job *j = (batch_job *)data;
  napi_value v;
  if (status == napi_ok && j->rc == 0) {
    void *out;
    napi_create_buffer_copy(env, 256 * j->count, j->proofs, &out, &v);
    napi_resolve_deferred(env, j->deferred, v);
  } else {
    napi_value msg;
    napi_create_string_utf8(env, j->rc ? j->err : "proof job cancelled", NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, NULL, msg, &v);
    napi_reject_deferred(env, j->deferred, v);
  }
  batch_free(env, j);
}
static int is_nullish(napi_env env, napi_value v) {
  napi_valuetype t;
  return napi_typeof(env, v, &t) == napi_ok && (t == napi_undefined || t == napi_null);
}
/* proveBatch(key, ...) and proveBatchMulti([key replicas...], ...) share everything but their first argument */
static napi_value prove_batch_common(napi_env env, napi_callback_info info, int multi) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (!Z.handle) return throw_msg(env, "libzkr_hip.so not loaded");
  if (argc < 2) return throw_msg(env, "proveBatch(key, witnessBins, rs, ss) / proveBatchMulti(keys, witnessBins, rs, ss)");
  batch_job *j = (batch_job *)calloc(1, sizeof(batch_job));
  if (!j) return throw_msg(env, "proveBatch: out of memory");
  uint32_t count = 0;
  if (multi) {
    uint32_t nk = 0;
    if (napi_get_array_length(env, argv[0], &nk) != napi_ok || nk == 0 || !(j->keys = (zkr_key **)calloc(nk, sizeof(zkr_key *)))) {
      batch_free(env, j);
      return throw_msg(env, "proveBatchMulti: a non-empty array of key handles expected");
    }
    j->n_keys = nk;
    for (uint32_t i = 0; i < nk; i++) {
      napi_value kv;
      if (napi_get_element(env, argv[0], i, &kv) != napi_ok || napi_get_value_external(env, kv, (void **)&j->keys[i]) != napi_ok) {
        batch_free(env, j);
        return throw_msg(env, "proveBatchMulti: every entry of keys must be a key handle");
      }
    }
  } else if (napi_get_value_external(env, argv[0], (void **)&j->key) != napi_ok) {
    batch_free(env, j);
    return throw_msg(env, "proveBatch: a key and a non-empty array of witness buffers expected");
  }
  if (napi_get_array_length(env, argv[1], &count) != napi_ok || count == 0) {
    batch_free(env, j);
    return throw_msg(env, "proveBatch: a key and a non-empty array of witness buffers expected");
  }
  j->count = count;
  j->wits = (const void **)calloc(count, sizeof(void *));
  j->wit_refs = (napi_ref *)calloc(count, sizeof(napi_ref));
  j->proofs = (uint8_t *)malloc(256 * (size_t)count);
  if (!j->wits || !j->wit_refs || !j->proofs) { batch_free(env, j); return throw_msg(env, "proveBatch: out of memory"); }
  for (uint32_t i = 0; i < count; i++) {
    napi_value w;
    const uint8_t *p;
    size_t n;
    if (napi_get_element(env, argv[1], i, &w) != napi_ok || !get_bytes(env, w, &p, &n) || (i && n != j->wit_len)) {
      batch_free(env, j);
      return throw_msg(env, "proveBatch: every witness must be an ArrayBuffer / Buffer / TypedArray of the same length");
    }
    if (napi_create_reference(env, w, 1, &j->wit_refs[i]) != napi_ok) { batch_free(env, j); return throw_msg(env, "proveBatch: cannot pin a witness buffer"); }
    j->n_refs = i + 1;  /* pinned before its pointer is kept */
    j->wits[i] = p;
    j->wit_len = n;
  }
  /* blinding: both rs and ss (32 bytes per proof each), or neither (null / undefined / omitted: drawn per proof).  Anything
   * else is an error -- a buffer of the wrong length must not silently turn into random blinding. */
  const int have_rs = argc >= 3 && !is_nullish(env, argv[2]), have_ss = argc >= 4 && !is_nullish(env, argv[3]);
  if (have_rs || have_ss) {
    const uint8_t *pr, *ps;
    size_t nr, ns;
    if (!have_rs || !have_ss || !get_bytes(env, argv[2], &pr, &nr) || !get_bytes(env, argv[3], &ps, &ns) || nr != 32 * (size_t)count || ns != 32 * (size_t)count) {
      batch_free(env, j);
      return throw_msg(env, "proveBatch: rs and ss must both hold 32 bytes per proof (or both be null)");
    }
    j->rs = (uint8_t *)malloc(nr);
    j->ss = (uint8_t *)malloc(ns);
    if (!j->rs || !j->ss) { batch_free(env, j); return throw_msg(env, "proveBatch: out of memory"); }
    memcpy(j->rs, pr, nr);
    memcpy(j->ss, ps, ns);
  }
  napi_value promise, name;
  if (napi_create_promise(env, &j->deferred, &promise) != napi_ok || napi_create_reference(env, argv[0], 1, &j->key_ref) != napi_ok ||
      napi_create_string_utf8(env, "zkr_prove_batch", NAPI_AUTO_LENGTH, &name) != napi_ok ||
      napi_create_async_work(env, NULL, name, batch_execute, batch_complete, j, &j->work) != napi_ok || napi_queue_async_work(env, j->work) != napi_ok) {
    if (j->deferred) {  /* as in prove_common: a promise nobody will settle is rejected before the job goes */
      napi_value msg, err;
      if (napi_create_string_utf8(env, "proveBatch: could not queue the proof job", NAPI_AUTO_LENGTH, &msg) == napi_ok && napi_create_error(env, NULL, msg, &err) == napi_ok)
        napi_reject_deferred(env, j->deferred, err);
    }
    batch_free(env, j);  /* nothing was queued: the job and its references go here */
    return throw_msg(env, "proveBatch: could not queue the proof job");
  }
  return promise;
}
static napi_value js_prove_batch(napi_env env, napi_callback_info info) { return prove_batch_common(env, info, 0); }
/* proveBatchMulti([key replicas], [witnessBin...], rs|null, ss|null) -> Promise<Buffer(256 * count)>: zkr_prove_batch_multi on a
 * libuv worker, which runs one host thread per replica (proof i on keys[i mod n]); the array of handles is referenced until
 * the promise settles, so no replica is finalized under the job. */
static napi_value js_prove_batch_multi(napi_env env, napi_callback_info info) { return prove_batch_common(env, info, 1); }

/* verify(vkBin, proofBytes256, publicBytes) -> boolean: the Groth16 pairing check on the host (zkr_verify; groth.isValid,
 * operator/src/snarks/common.ts:30-34).  Synchronous: a few milliseconds, no GPU. */
static napi_value js_verify(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value argv[3];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (!Z.handle) return throw_msg(env, "libzkr_hip.so not loaded");
  const uint8_t *vk, *proof, *pub;
  size_t vk_len, proof_len, pub_len;
  if (argc < 3 || !get_bytes(env, argv[0], &vk, &vk_len) || !get_bytes(env, argv[1], &proof, &proof_len) || !get_bytes(env, argv[2], &pub, &pub_len))
    return throw_msg(env, "verify(vkBin, proofBytes, publicBytes): three byte buffers expected");
  if (proof_len != 256 || pub_len % 32) return throw_msg(env, "verify: proof must be 256 bytes and the inputs a multiple of 32 bytes");
  int ok = 0;
  if (Z.verify(vk, vk_len, proof, pub, pub_len / 32, &ok)) return throw_msg(env, Z.last_error());
  napi_value out;
  NAPI_OK(napi_get_boolean(env, ok != 0, &out));
  return out;
}

/* ---- the rollup circuit without circom / snarkjs (include/zkr.h); all synchronous host work ---- */
/* rollupCrypto(op, a, b, c): op 0 multiHash(values) -> 32 B; 1 pubKey(priv32) -> 64 B; 2 sign(priv32, msg) -> 96 B;
 * 3 verify(msg, sig96, pub64) -> boolean; 4 formatPrivKeyForBabyJub(priv32) -> 32 B.  Field elements are 32 B little-endian (operator/src/utils/crypto.ts). */
static napi_value js_rollup_crypto(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (!Z.handle) return throw_msg(env, "libzkr_hip.so not loaded");
  int32_t op = -1;
  const uint8_t *a = NULL, *b = NULL, *c = NULL;
  size_t al = 0, bl = 0, cl = 0;
  if (argc < 2 || napi_get_value_int32(env, argv[0], &op) != napi_ok || !get_bytes(env, argv[1], &a, &al)) return throw_msg(env, "rollupCrypto(op, bytes, ...)");
  if (argc > 2 && !get_bytes(env, argv[2], &b, &bl)) return throw_msg(env, "rollupCrypto: byte buffer expected");
  if (argc > 3 && !get_bytes(env, argv[3], &c, &cl)) return throw_msg(env, "rollupCrypto: byte buffer expected");
  uint8_t out[96];
  size_t out_len = 0;
  int rc, ok = 0;
  napi_value res;
  switch (op) {
    case 0:
      if (al % 32) return throw_msg(env, "multiHash: values are 32 bytes each");
      rc = Z.multihash(a, al / 32, out), out_len = 32;
      break;
    case 1:
      if (al != 32) return throw_msg(env, "genPublicKey: the private key is 32 bytes");
      rc = Z.pubkey(a, out), out_len = 64;
      break;
    case 2:
      if (al != 32 || bl % 32) return throw_msg(env, "sign: 32-byte private key, 32 bytes per message element");
      rc = Z.eddsa_sign(a, b, bl / 32, out), out_len = 96;
      break;
    case 3:
      if (al % 32 || bl != 96 || cl != 64) return throw_msg(env, "verify: message elements of 32 bytes, 96-byte signature, 64-byte public key");
      rc = Z.eddsa_verify(a, al / 32, b, c, &ok);
      if (rc) return throw_msg(env, Z.last_error());
      NAPI_OK(napi_get_boolean(env, ok != 0, &res));
      return res;
    case 4:
      if (al != 32) return throw_msg(env, "formatPrivKeyForBabyJub: the private key is 32 bytes");
      rc = Z.format_privkey(a, out), out_len = 32;
      break;
    default:
      return throw_msg(env, "rollupCrypto: unknown operation");
  }
  if (rc) return throw_msg(env, Z.last_error());
  void *copy;
  NAPI_OK(napi_create_buffer_copy(env, out_len, out, &copy, &res));
  return res;
}

/* rollupCircuit(batch, depth, inputs|null): inputs == null -> [nVars, nPublic, nConstraints, r1csBin Buffer];
 * inputs = (nPublic - 1) x 32 B -> the witness as binarifyWitness lays it out (Buffer); throws where
 * Circuit.calculateWitness would (operator/src/snarks/common.ts:15-17). */
static napi_value js_rollup_circuit(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value argv[3];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (!Z.handle) return throw_msg(env, "libzkr_hip.so not loaded");
  uint32_t batch = 0, depth = 0;
  if (argc < 2 || napi_get_value_uint32(env, argv[0], &batch) != napi_ok || napi_get_value_uint32(env, argv[1], &depth) != napi_ok)
    return throw_msg(env, "rollupCircuit(batch, depth, inputs|null)");
  napi_valuetype t = napi_null;
  if (argc > 2) napi_typeof(env, argv[2], &t);
  void *blob = NULL, *copy;
  size_t len = 0;
  napi_value res, buf;
  if (t == napi_null || t == napi_undefined) {
    uint32_t v[3];
    if (Z.rollup_info(batch, depth, &v[0], &v[1], &v[2]) || Z.rollup_r1cs(batch, depth, &blob, &len)) return throw_msg(env, Z.last_error());
    NAPI_OK(napi_create_array_with_length(env, 4, &res));
    for (int i = 0; i < 3; i++) {
      napi_value n;
      NAPI_OK(napi_create_uint32(env, v[i], &n));
      NAPI_OK(napi_set_element(env, res, i, n));
    }
    napi_status st = napi_create_buffer_copy(env, len, blob, &copy, &buf);
    Z.free_(blob);
    if (st != napi_ok) return throw_msg(env, "zkr_napi: cannot allocate the R1CS buffer");
    NAPI_OK(napi_set_element(env, res, 3, buf));
    return res;
  }
  const uint8_t *in;
  size_t in_len;
  if (!get_bytes(env, argv[2], &in, &in_len) || in_len % 32) return throw_msg(env, "rollupCircuit: inputs are 32 bytes each");
  if (Z.rollup_witness(batch, depth, in, in_len / 32, &blob, &len)) return throw_msg(env, Z.last_error());
  napi_status st = napi_create_buffer_copy(env, len, blob, &copy, &buf);
  Z.free_(blob);
  if (st != napi_ok) return throw_msg(env, "zkr_napi: cannot allocate the witness buffer");
  return buf;
}

/* rollupGpuHash(values, arity, device) -> Buffer of one 32-byte hash per `arity` values (zkr_mimcsponge_multihash_batch);
 * arity 0: `values` are the 2^depth leaves of a balance tree, the result is every level, leaves first, root last
 * (zkr_balance_tree_build).  Synchronous; runs on the GPU. */
static napi_value js_rollup_gpu_hash(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value argv[3];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (!Z.handle) return throw_msg(env, "libzkr_hip.so not loaded");
  const uint8_t *in;
  size_t len;
  uint32_t arity = 0;
  int32_t dev = 0;
  if (argc < 2 || !get_bytes(env, argv[0], &in, &len) || len % 32 || napi_get_value_uint32(env, argv[1], &arity) != napi_ok)
    return throw_msg(env, "rollupGpuHash(values, arity, device)");
  if (argc > 2) napi_get_value_int32(env, argv[2], &dev);
  size_t n = len / 32, out_len;
  unsigned depth = 0;
  if (arity == 0) {
    while (((size_t)1 << depth) < n) depth++;
    if (n < 2 || ((size_t)1 << depth) != n) return throw_msg(env, "balance tree: the number of leaves must be a power of two");
    out_len = 32 * (2 * n - 1);
  } else {
    if (n % arity) return throw_msg(env, "multiHashBatch: rows of unequal length");
    out_len = 32 * (n / arity);
  }
  void *out = NULL;
  napi_value buf;
  NAPI_OK(napi_create_buffer(env, out_len, &out, &buf));
  int rc = arity == 0 ? Z.tree_build(in, depth, out, dev) : Z.multihash_batch(in, n / arity, arity, out, dev);
  if (rc) return throw_msg(env, Z.last_error());
  return buf;
}

/* withdrawCircuit(privateKey32|null, nullifier32): null -> r1csBin Buffer; else the witness (Buffer) of Withdraw()
 * (prover/circuits/withdraw.circom:4-25). */
static napi_value js_withdraw_circuit(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (!Z.handle) return throw_msg(env, "libzkr_hip.so not loaded");
  napi_valuetype t = napi_null;
  if (argc > 0) napi_typeof(env, argv[0], &t);
  void *blob = NULL, *copy;
  size_t len = 0;
  napi_value buf;
  if (t == napi_null || t == napi_undefined) {
    if (Z.withdraw_r1cs(&blob, &len)) return throw_msg(env, Z.last_error());
  } else {
    const uint8_t *k, *nul;
    size_t kl, nl;
    if (argc < 2 || !get_bytes(env, argv[0], &k, &kl) || !get_bytes(env, argv[1], &nul, &nl) || kl != 32 || nl != 32)
      return throw_msg(env, "withdrawCircuit(privateKey32, nullifier32)");
    if (Z.withdraw_witness(k, nul, &blob, &len)) return throw_msg(env, Z.last_error());
  }
  napi_status st = napi_create_buffer_copy(env, len, blob, &copy, &buf);
  Z.free_(blob);
  if (st != napi_ok) return throw_msg(env, "zkr_napi: cannot allocate the result buffer");
  return buf;
}

/* verifyBatch(vkBin, proofs (n x 256 B), publics (n x nPublic x 32 B), n) -> boolean: every proof verifies (zkr_verify_batch) */
static napi_value js_verify_batch(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (!Z.handle) return throw_msg(env, "libzkr_hip.so not loaded");
  const uint8_t *vk, *proofs, *pub;
  size_t vk_len, proofs_len, pub_len;
  uint32_t n = 0;
  if (argc < 4 || !get_bytes(env, argv[0], &vk, &vk_len) || !get_bytes(env, argv[1], &proofs, &proofs_len) || !get_bytes(env, argv[2], &pub, &pub_len) ||
      napi_get_value_uint32(env, argv[3], &n) != napi_ok)
    return throw_msg(env, "verifyBatch(vkBin, proofs, publics, n)");
  if (n == 0 || proofs_len != 256 * (size_t)n || pub_len % (32 * (size_t)n)) return throw_msg(env, "verifyBatch: n proofs of 256 bytes and n equal rows of 32-byte inputs expected");
  int ok = 0;
  if (Z.verify_batch(vk, vk_len, proofs, pub, n, pub_len / 32 / n, &ok)) return throw_msg(env, Z.last_error());
  napi_value out;
  NAPI_OK(napi_get_boolean(env, ok != 0, &out));
  return out;
}

static napi_value init(napi_env env, napi_value exports) {
  napi_property_descriptor props[] = {
      {"load", NULL, js_load, NULL, NULL, NULL, napi_default, NULL},       {"version", NULL, js_version, NULL, NULL, NULL, napi_default, NULL},
      {"keyLoad", NULL, js_key_load, NULL, NULL, NULL, napi_default, NULL}, {"keyInfo", NULL, js_key_info, NULL, NULL, NULL, napi_default, NULL},
      {"prove", NULL, js_prove, NULL, NULL, NULL, napi_default, NULL},     {"verify", NULL, js_verify, NULL, NULL, NULL, napi_default, NULL},
      {"proveBatch", NULL, js_prove_batch, NULL, NULL, NULL, napi_default, NULL},
      {"proveBatchMulti", NULL, js_prove_batch_multi, NULL, NULL, NULL, napi_default, NULL},
      {"keyShard", NULL, js_key_shard, NULL, NULL, NULL, napi_default, NULL}, {"proveSharded", NULL, js_prove_sharded, NULL, NULL, NULL, napi_default, NULL},
      {"keyReplicate", NULL, js_key_replicate, NULL, NULL, NULL, napi_default, NULL}, {"keyDevice", NULL, js_key_device, NULL, NULL, NULL, napi_default, NULL},
      {"setupR1cs", NULL, js_setup_r1cs, NULL, NULL, NULL, napi_default, NULL}, {"keySave", NULL, js_key_save, NULL, NULL, NULL, napi_default, NULL},
      {"keyLoadFile", NULL, js_key_load_file, NULL, NULL, NULL, napi_default, NULL},
      {"rollupCrypto", NULL, js_rollup_crypto, NULL, NULL, NULL, napi_default, NULL}, {"rollupCircuit", NULL, js_rollup_circuit, NULL, NULL, NULL, napi_default, NULL},
      {"withdrawCircuit", NULL, js_withdraw_circuit, NULL, NULL, NULL, napi_default, NULL},
      {"rollupGpuHash", NULL, js_rollup_gpu_hash, NULL, NULL, NULL, napi_default, NULL},
      {"verifyBatch", NULL, js_verify_batch, NULL, NULL, NULL, napi_default, NULL},
      {"shardedLastForm", NULL, js_sharded_last_form, NULL, NULL, NULL, napi_default, NULL},
      {"keyReplication", NULL, js_key_replication, NULL, NULL, NULL, napi_default, NULL},
  };
  napi_define_properties(env, exports, sizeof(props) / sizeof(props[0]), props);
  return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
