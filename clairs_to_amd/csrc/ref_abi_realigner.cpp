// The reference's native realigner ABI, symbol for symbol, on top of libclairsto_amd.so - so that
//   realigner = ctypes.cdll.LoadLibrary(realigner_mod)            (src/realign_reads.py:70)
//   realigner.realign_reads(seq_list, position_list, cigars_list, ref_seq, haplotypes, tmp_ref_start, len(ref_prefix),
//                           len(ref_suffix), total_read_num) -> POINTER(StructPointer)      (:582-591)
//   realigner.free_memory(realigner_p, total_read_num)            (:613-615)
// work unchanged when `realigner_mod` points at clairs_to_amd/realign/realigner.so.  Layout of the result = `struct_str_arr`
// of src/realign/realigner.h:42-46 = the caller's `StructPointer` (:74-77): int position[1000]; char* cigar_string[1000].
// A failing call (CTO_EINVAL, see include/clairsto_amd.h) reports on stderr and hands every read back unchanged.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/clairsto_amd.h"

namespace { constexpr int kMaxReads = 1000; }
struct cto_ref_realign_out { int position[kMaxReads]; char* cigar_string[kMaxReads]; };

extern "C" cto_ref_realign_out* realign_reads(char* seqs[], int* positions, char* cigars[], char* reference, char* haplotypes,
                                              int ref_start, int ref_prefix, int ref_suffix, int read_size) {
    cto_ref_realign_out* out = static_cast<cto_ref_realign_out*>(calloc(1, sizeof(cto_ref_realign_out)));
    if (!out || read_size < 0 || read_size > kMaxReads) return out;
    size_t cap = 64;
    for (int i = 0; i < read_size; ++i) cap += 8 * strlen(seqs[i]) + strlen(cigars[i]) + 64;
    std::vector<char> buf(cap);
    std::vector<int64_t> off(read_size + 1);
    std::vector<int32_t> pos(positions, positions + read_size), new_pos(read_size);
    const int rc = cto_realign_reads(read_size, seqs, pos.data(), cigars, reference, haplotypes, ref_start, ref_prefix, ref_suffix,
                                     new_pos.data(), buf.data(), cap, off.data());
    if (rc != CTO_OK) fprintf(stderr, "[clairs_to_amd] realign_reads: %s\n", cto_last_error());
    for (int i = 0; i < read_size; ++i) {
        out->position[i] = rc == CTO_OK ? new_pos[i] : positions[i];
        out->cigar_string[i] = strdup(rc == CTO_OK ? buf.data() + off[i] : cigars[i]);
    }
    return out;
}

extern "C" void free_memory(cto_ref_realign_out* p, int size) {
    if (!p) return;
    for (int i = 0; i < size && i < kMaxReads; ++i) free(p->cigar_string[i]);
    free(p);
}
