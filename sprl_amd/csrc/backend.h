// backend.h — the seam between the engine host code and the device runtime.
// The product implements it with HIP for gfx950 (backend_hip.hip).  tests/emu implements it on the CPU SIMT
// emulator for kernel-logic checks; that build is never shipped and the product has no fallback path.
#ifndef SPRL_BACKEND_H
#define SPRL_BACKEND_H

#include <stddef.h>

#include <string>

#include "engine_types.h"

struct RecPacked;       // records_kernel.h
struct RecExpanded;

namespace be {
bool available(std::string* why);
int init(int device, std::string* err);
void* dmalloc(size_t bytes);
void dfree(void* p);
int h2d(void* dst, const void* src, size_t n);
int d2h(void* dst, const void* src, size_t n);
int dmemset(void* dst, int v, size_t n);
int sync();
// every operation above and below goes to the calling thread's current stream (null stream unless set)
void* stream_create();
void stream_destroy(void* s);
void set_stream(void* s);
void bind(int device, void* s);   // make `device` current on the calling thread and `s` its stream
void* current_stream();
// lab (SPRL_TREE_STREAM): a second stream per engine for the tree / scan / gather launches, at the highest priority the device
// offers when `high`, tied to the engine's stream by two re-recorded events
void* stream_create_priority(int high);
void* event_new();
void event_free(void* ev);
void event_record(void* ev, void* stream);
void stream_wait(void* stream, void* ev);
// one 64-lane wavefront per game slot, on the null stream
int launch_step(int game, const EngineParams& P);
// match play: slots (pair, pair + num_slots/2) hold the two agents' trees of one game; Othello, Connect Four, Go 7x7
int launch_match(int game, const EngineParams& P);
// leaf_count -> leaf_offset (exclusive scan, slot order) + counters->leaf_total, then gather the queued leaves'
// planes from the sparse staging into the dense network batch
int launch_compact(const EngineParams& P, int floats_per_leaf);
// finished-game records on the device (records_kernel.h): offsets scan, wire-format packing, training-sample expansion
int launch_records_scan(const EngineParams& P);
int launch_records_pack(int game, const EngineParams& P, const RecPacked& out, int use_sym);
int launch_records_expand(int game, const EngineParams& P, const RecExpanded& out);
// event pairs on the null stream (profile mode); returns milliseconds between the two marks
void* mark();
double elapsed_ms(void* a, void* b);   // synchronises on b
void mark_free(void* m);
// profile mode, tree kernel: the pair's duration (synchronises on b), the interval also logged on the process-wide clock of
// busy_log.h (all engines of the process: bench.py --populations); both marks are consumed.  `chain` from chain_new().
void* chain_new();
void chain_free(void* chain);
double resolve_logged(void* chain, void* a, void* b);
double busy_ms(double* sum_ms);     // time with >= 1 tree kernel executing since busy_reset(); sum_ms = sum of the durations
void busy_reset();
const char* name();
const char* last_error();
}  // namespace be

#endif
