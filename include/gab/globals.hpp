// globals.hpp — CLI-backed process globals and the legacy result writers
// (reference: cuda/globals.cuh:20-42, cuda/globals.cu).  Names and defaults are
// the reference's; the CSV / JSON formats are byte-compatible with it.
#pragma once
// cuda/globals.cuh:37 (a Windows path there; nothing reads it)
#define OUTFILE "/tmp/latencies.txt"


#include <string>
#include <vector>

extern int NTRACKS;              // --nTracks      (128)
extern int FS;                   // --fs           (48000)
extern int BUFSIZE;              // --bufferSize   (512)
extern int NRUNS;                // --nRuns        (100)
extern std::string OUTPUT_FILE;  // --outputfile   ("")
extern bool JSON_OUTPUT;         // --json

// Additions of the MI355X build (unreachable from the reference CLI):
extern int IR_LENGTH;            // --irLength     (<=0: each benchmark's DEFAULT_IR_LEN)
extern int FDTD_GRID;            // --fdtdGrid     (<=0: 52, the reference's 50+2)
extern int CONV_STREAMING;       // --convMode stateless (0) | stream (1, default) | roundtrip (2: stream, overlapped iteration)
extern int MODAL_REAL;           // --modalMode placeholder|bank (default placeholder = the CUDA port)
extern bool GAB_QUIET;           // suppress progress chatter (library use)

// DAW-simulation knobs exist in the reference as compile-time macros, all off
// (cuda/globals.cuh:28-30).
#define ENABLE_DAWSIM_SLEEP false
#define SLEEP_MS 90
#define ENABLE_DAWSIM_SPIN false

void writeVectorToFile(const std::vector<float>& vec, const std::string& filename);
void printVectorStats(const std::vector<float>& vec);
void writeCSVResults(const std::vector<float>& vec, const std::string& benchmarkName,
                     const std::string& filename);
void writeJSONResults(const std::vector<float>& vec, const std::string& benchmarkName,
                      const std::string& filename = "");
std::string generateJSONResults(const std::vector<float>& vec, const std::string& benchmarkName);
// The reference's object with further members appended before its closing brace (additive):
// `extra_members` is the text of one or more `"key": value` members, comma-separated, no trailing comma.
std::string generateJSONResultsWith(const std::vector<float>& vec, const std::string& benchmarkName,
                                    const std::string& extra_members);
extern int CONV_BATCH;           // --convBatch    (<=1: one buffer per iteration with its H2D / D2H copies, as the reference;
                                 //                 n: n HBM-resident buffers per iteration in ONE gab_conv_process_batch launch)
extern int FDTD_FORM;            // --fdtdForm     (0 auto: LDS-resident where the room fits, 1 step: one launch per step)
extern int DATACOPY_SEQUENTIAL;  // --datacopyMode (0 overlap: upload and download at once, gab_datatransfer_round_trip;
                                 //                 1 sequential: H2D -> kernel -> D2H on one stream, as the reference)
extern int FDTD_STEPS;           // --fdtdSteps    (<=0: bufferSize samples x 3 steps, as the reference)
extern int CPU_THREADS;          // --cpu-threads  (<=0: every hardware thread) for the timed CPU golden
