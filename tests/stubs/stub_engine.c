/* stub_engine.c — a stand-in for libmi355_llama.so's engine entry points (include/mi355_llama.h, "engine" section) so that the HTTP host's
 * transport and relaying can be tested without a GPU: canned JSON, and a "streaming completion" delivered from another thread after the call
 * has returned (as the real engine does).  Test infrastructure only. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

typedef void (*cb_t)(const char *status_json, const char *body_json, void *user);
static volatile int g_stop = 0;
static const char *OK = "{\"is_done\":true,\"has_error\":false,\"is_stream\":false,\"status_code\":200}";
static const char *E409 = "{\"is_done\":true,\"has_error\":true,\"is_stream\":false,\"status_code\":409}";
static const char *E500 = "{\"is_done\":true,\"has_error\":true,\"is_stream\":false,\"status_code\":500}";
static const char *NOT_LOADED = "{\"message\":\"Model is not loaded yet\"}";

void *mi355_engine_create(void) { return malloc(8); }
void mi355_engine_destroy(void *e) { free(e); }
const char *mi355_last_error(void) { return ""; }
void mi355_engine_load_model(void *e, const char *b, cb_t cb, void *u) {
    (void)e;
    if (strstr(b, "nonexistent")) cb(E500, "{\"message\":\"Failed to load model\"}", u);
    else cb(OK, "{\"message\":\"Model loaded successfully\"}", u);
}
void mi355_engine_unload_model(void *e, const char *b, cb_t cb, void *u) { (void)e; (void)b; cb(E409, NOT_LOADED, u); }
void mi355_engine_get_model_status(void *e, const char *b, cb_t cb, void *u) { (void)e; (void)b; cb(E409, NOT_LOADED, u); }
void mi355_engine_get_models(void *e, const char *b, cb_t cb, void *u) { (void)e; (void)b; cb(OK, "{\"object\":\"list\",\"data\":[]}", u); }
void mi355_engine_handle_embedding(void *e, const char *b, cb_t cb, void *u) { (void)e; (void)b; cb(E409, NOT_LOADED, u); }
void mi355_engine_stop_inferencing(void *e, const char *model) {
    (void)e;
    g_stop = 1;
    const char *f = getenv("STUB_ENGINE_STOP_FILE");
    if (f) { FILE *o = fopen(f, "w"); if (o) { fprintf(o, "%s", model); fclose(o); } }
}

struct job { cb_t cb; void *u; int n; };
static void *stream_thread(void *p) {
    struct job *j = (struct job *)p;
    char body[256];
    for (int i = 0; i < j->n && !g_stop; i++) {
        usleep(5000);
        snprintf(body, sizeof body, "{\"data\":\"data: {\\\"object\\\":\\\"chat.completion.chunk\\\",\\\"i\\\":%d}\\n\\n\"}", i);
        j->cb("{\"is_done\":false,\"has_error\":false,\"is_stream\":true,\"status_code\":200}", body, j->u);
    }
    j->cb("{\"is_done\":true,\"has_error\":false,\"is_stream\":true,\"status_code\":200}", "{\"data\":\"data: [DONE]\\n\\n\"}", j->u);
    free(j);
    return NULL;
}
void mi355_engine_handle_chat_completion(void *e, const char *b, cb_t cb, void *u) {
    (void)e;
    if (strstr(b, "\"nope\"")) { cb(E409, NOT_LOADED, u); return; }
    if (strstr(b, "\"stream\": true") || strstr(b, "\"stream\":true")) {
        struct job *j = (struct job *)malloc(sizeof *j);
        j->cb = cb; j->u = u; j->n = strstr(b, "\"long\"") ? 2000 : 5;
        g_stop = 0;
        pthread_t t;
        pthread_create(&t, NULL, stream_thread, j);
        pthread_detach(t);
        return;
    }
    cb(OK, "{\"object\":\"chat.completion\",\"choices\":[{\"message\":{\"role\":\"assistant\",\"content\":\"stub\"}}]}", u);
}
