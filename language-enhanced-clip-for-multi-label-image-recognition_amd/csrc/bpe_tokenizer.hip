// Native byte-level BPE tokenizer (host code; SURVEY.md §8f N4 "native BPE"): the arithmetic of the reference's
// clip/simple_tokenizer.py:62-132 (SimpleTokenizer.bpe / .encode) and clip/clip.py:185-221 (tokenize) in C++ behind the C ABI.
//   text -> whitespace collapse + strip + lower-case -> the CLIP split pattern
//           <|startoftext|> | <|endoftext|> | 's|'t|'re|'ve|'m|'ll|'d | \p{L}+ | \p{N} | [^\s\p{L}\p{N}]+
//        -> UTF-8 bytes through the GPT-2 byte alphabet -> greedy lowest-rank pair merging (</w> on the last symbol) -> ids
// Integer / byte work: the ids are bit-exact or wrong.  Unicode classes and lower-casing come from tables generated from the `regex`
// module and Python's str.lower() (csrc/tools/gen_unicode_tables.py), so any UTF-8 text is classified exactly as the reference's
// regex does.  Out of its scope (the caller keeps them on the Python side, as the reference does with ftfy / html): html entity
// unescaping - a text containing '&' is refused with LECLIP_E_UNSUPPORTED - and ftfy's mojibake repair.
// The merge table is the checkpoint-side data file bpe_simple_vocab_16e6.txt.gz (read through zlib); it is not vendored.
#include <stdint.h>
#include <string.h>
#include <zlib.h>

#include <mutex>
#include <shared_mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/leclip_hip.h"
#include "unicode_tables.h"

void leclip_set_error(const char* fmt, ...);

namespace {

constexpr int N_MERGES = 49152 - 256 - 2;

bool in_ranges(const unsigned (*r)[2], int n, unsigned cp) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        if (cp < r[mid][0]) hi = mid - 1;
        else if (cp > r[mid][1]) lo = mid + 1;
        else return true;
    }
    return false;
}
inline bool is_L(unsigned cp) { return in_ranges(UNI_L, UNI_L_count, cp); }
inline bool is_N(unsigned cp) { return in_ranges(UNI_N, UNI_N_count, cp); }
inline bool is_S(unsigned cp) { return in_ranges(UNI_S, UNI_S_count, cp); }

void put_utf8(std::string& s, unsigned cp) {
    if (cp < 0x80) s.push_back((char)cp);
    else if (cp < 0x800) { s.push_back((char)(0xC0 | (cp >> 6))); s.push_back((char)(0x80 | (cp & 63))); }
    else if (cp < 0x10000) { s.push_back((char)(0xE0 | (cp >> 12))); s.push_back((char)(0x80 | ((cp >> 6) & 63))); s.push_back((char)(0x80 | (cp & 63))); }
    else { s.push_back((char)(0xF0 | (cp >> 18))); s.push_back((char)(0x80 | ((cp >> 12) & 63))); s.push_back((char)(0x80 | ((cp >> 6) & 63))); s.push_back((char)(0x80 | (cp & 63))); }
}

// strict UTF-8 decode; returns false on malformed input
bool decode_utf8(const char* s, std::vector<unsigned>& out) {
    const unsigned char* p = (const unsigned char*)s;
    while (*p) {
        unsigned cp;
        int n;
        if (*p < 0x80) { cp = *p; n = 1; }
        else if ((*p & 0xE0) == 0xC0) { cp = *p & 0x1F; n = 2; }
        else if ((*p & 0xF0) == 0xE0) { cp = *p & 0x0F; n = 3; }
        else if ((*p & 0xF8) == 0xF0) { cp = *p & 0x07; n = 4; }
        else return false;
        for (int i = 1; i < n; ++i) {
            if ((p[i] & 0xC0) != 0x80) return false;
            cp = (cp << 6) | (p[i] & 63);
        }
        out.push_back(cp);
        p += n;
    }
    return true;
}

struct Bpe {
    std::unordered_map<std::string, int> encoder;       // token -> id
    std::unordered_map<std::string, int> rank;          // "a\x01b" -> merge rank
    std::string byte_sym[256];                          // GPT-2 byte alphabet (UTF-8 of the mapped code point)
    // word -> ids memo.  The handle is shared process-wide (clip.py keeps one) and ctypes drops the GIL during a call, so two
    // Python threads (a threaded caption loader) may encode at once: lookups take the lock shared, inserts exclusive.  The
    // other members are read-only after leclip_bpe_open.
    std::unordered_map<std::string, std::vector<int>> cache;
    std::shared_mutex cache_mu;
    int sot, eot;
};

void build_alphabet(Bpe& b, std::vector<std::string>& order) {
    bool keep[256] = {};
    for (int c = 33; c <= 126; ++c) keep[c] = true;
    for (int c = 161; c <= 172; ++c) keep[c] = true;
    for (int c = 174; c <= 255; ++c) keep[c] = true;
    int extra = 0;
    for (int c = 0; c < 256; ++c) {
        std::string s;
        put_utf8(s, keep[c] ? (unsigned)c : 256u + extra++);
        b.byte_sym[c] = s;
    }
    // vocabulary order: kept bytes in increasing order, then the remapped ones (simple_tokenizer.py:18-34, 68-69)
    for (int c = 0; c < 256; ++c) if (keep[c]) order.push_back(b.byte_sym[c]);
    for (int c = 0; c < 256; ++c) if (!keep[c]) order.push_back(b.byte_sym[c]);
}

bool read_gz(const char* path, std::string& out) {
    gzFile f = gzopen(path, "rb");
    if (!f) return false;
    char buf[65536];
    int n;
    while ((n = gzread(f, buf, sizeof(buf))) > 0) out.append(buf, n);
    gzclose(f);
    return n == 0;
}

// greedy BPE on one pre-token (already mapped through the byte alphabet): ids appended to `ids`
void bpe_word(Bpe& b, const std::vector<std::string>& syms_in, const std::string& key, std::vector<int>& ids) {
    {
        std::shared_lock<std::shared_mutex> rd(b.cache_mu);
        auto hit = b.cache.find(key);
        if (hit != b.cache.end()) { ids.insert(ids.end(), hit->second.begin(), hit->second.end()); return; }
    }
    std::vector<std::string> parts = syms_in;
    parts.back() += "</w>";
    while (parts.size() > 1) {
        int best = -1, best_rank = 0;
        for (size_t i = 0; i + 1 < parts.size(); ++i) {
            auto r = b.rank.find(parts[i] + '\x01' + parts[i + 1]);
            if (r != b.rank.end() && (best < 0 || r->second < best_rank)) { best = (int)i; best_rank = r->second; }
        }
        if (best < 0) break;
        const std::string a = parts[best], c = parts[best + 1];
        std::vector<std::string> fused;
        for (size_t i = 0; i < parts.size();) {
            if (i + 1 < parts.size() && parts[i] == a && parts[i + 1] == c) { fused.push_back(a + c); i += 2; }
            else { fused.push_back(parts[i]); i += 1; }
        }
        parts.swap(fused);
    }
    std::vector<int> out;
    for (auto& p : parts) {
        auto e = b.encoder.find(p);
        out.push_back(e == b.encoder.end() ? -1 : e->second);
    }
    {
        std::unique_lock<std::shared_mutex> wr(b.cache_mu);
        b.cache.emplace(key, out);   // (a racing thread may have inserted the same word: identical value, emplace keeps the first)
    }
    ids.insert(ids.end(), out.begin(), out.end());
}

bool starts_with(const std::vector<unsigned>& t, size_t i, const char* lit) {
    size_t n = strlen(lit);
    if (i + n > t.size()) return false;
    for (size_t k = 0; k < n; ++k) if (t[i + k] != (unsigned char)lit[k]) return false;
    return true;
}

int encode_text(Bpe& b, const char* utf8, std::vector<int>& ids) {
    if (strchr(utf8, '&')) {
        leclip_set_error("bpe: text contains '&' (html entities are unescaped on the Python side, simple_tokenizer.py:40-43)");
        return LECLIP_E_UNSUPPORTED;
    }
    std::vector<unsigned> raw;
    if (!decode_utf8(utf8, raw)) { leclip_set_error("bpe: malformed UTF-8"); return LECLIP_E_INVALID; }
    for (unsigned cp : raw)
        if (cp == 0x3A3 || cp == 0x17F) {   // capital sigma lower-cases by context (final-sigma rule); long s case-folds onto the 's contraction
            leclip_set_error("bpe: U+03A3 / U+017F need Python's context-sensitive case handling");
            return LECLIP_E_UNSUPPORTED;
        }
    // whitespace_clean + strip + lower (simple_tokenizer.py:46-49, 124)
    std::vector<unsigned> t;
    bool pending_space = false;
    for (unsigned cp : raw) {
        if (is_S(cp)) { pending_space = !t.empty(); continue; }
        if (pending_space) { t.push_back(' '); pending_space = false; }
        // str.lower(): table lookup (multi-code-point results kept)
        int lo = 0, hi = UNI_LOWER_count - 1, found = -1;
        while (lo <= hi) {
            const int mid = (lo + hi) >> 1;
            if (cp < UNI_LOWER[mid][0]) hi = mid - 1;
            else if (cp > UNI_LOWER[mid][0]) lo = mid + 1;
            else { found = mid; break; }
        }
        if (found < 0) t.push_back(cp);
        else for (int k = 1; k < 4; ++k) if (UNI_LOWER[found][k]) t.push_back(UNI_LOWER[found][k]);
    }
    static const char* kSpecial[2] = {"<|startoftext|>", "<|endoftext|>"};
    static const char* kContr[7] = {"'s", "'t", "'re", "'ve", "'m", "'ll", "'d"};
    size_t i = 0;
    while (i < t.size()) {
        size_t j = i;
        bool special = false;
        for (int k = 0; k < 2 && !special; ++k)
            if (starts_with(t, i, kSpecial[k])) { ids.push_back(k == 0 ? b.sot : b.eot); i += strlen(kSpecial[k]); special = true; }
        if (special) continue;
        for (int k = 0; k < 7 && j == i; ++k)
            if (starts_with(t, i, kContr[k])) j = i + strlen(kContr[k]);
        if (j == i) {
            if (is_L(t[i])) { while (j < t.size() && is_L(t[j])) ++j; }
            else if (is_N(t[i])) { j = i + 1; }
            else if (!is_S(t[i])) { while (j < t.size() && !is_S(t[j]) && !is_L(t[j]) && !is_N(t[j])) ++j; }
            else { ++i; continue; }     // whitespace between tokens
        }
        std::string bytes;
        for (size_t k = i; k < j; ++k) put_utf8(bytes, t[k]);
        std::vector<std::string> syms;
        std::string key;
        for (unsigned char ch : bytes) { syms.push_back(b.byte_sym[ch]); key += b.byte_sym[ch]; }
        bpe_word(b, syms, key, ids);
        i = j;
    }
    for (int v : ids)
        if (v < 0) { leclip_set_error("bpe: token outside the vocabulary"); return LECLIP_E_INVALID; }
    return LECLIP_OK;
}

}  // namespace

extern "C" void* leclip_bpe_open(const char* vocab_gz_path) {
    if (!vocab_gz_path) { leclip_set_error("bpe_open: null path"); return nullptr; }
    std::string text;
    if (!read_gz(vocab_gz_path, text)) { leclip_set_error("bpe_open: cannot read %s", vocab_gz_path); return nullptr; }
    Bpe* b = new Bpe;
    std::vector<std::string> vocab;
    build_alphabet(*b, vocab);
    const size_t n_sym = vocab.size();
    for (size_t i = 0; i < n_sym; ++i) vocab.push_back(vocab[i] + "</w>");
    // merges: lines 1 .. N_MERGES of the file, "a b"
    size_t pos = text.find('\n');
    int n = 0;
    while (pos != std::string::npos && n < N_MERGES) {
        const size_t next = text.find('\n', pos + 1);
        const std::string line = text.substr(pos + 1, (next == std::string::npos ? text.size() : next) - pos - 1);
        pos = next;
        const size_t sp = line.find(' ');
        if (sp == std::string::npos || sp == 0 || sp + 1 >= line.size()) { if (line.empty()) continue; break; }
        const std::string a = line.substr(0, sp), c = line.substr(sp + 1);
        b->rank.emplace(a + '\x01' + c, n++);
        vocab.push_back(a + c);
    }
    if (n == 0) { leclip_set_error("bpe_open: %s holds no merges", vocab_gz_path); delete b; return nullptr; }
    vocab.push_back("<|startoftext|>");
    vocab.push_back("<|endoftext|>");
    for (size_t i = 0; i < vocab.size(); ++i) b->encoder.emplace(vocab[i], (int)i);
    b->sot = (int)vocab.size() - 2;
    b->eot = (int)vocab.size() - 1;
    return b;
}

extern "C" void leclip_bpe_close(void* handle) { delete (Bpe*)handle; }

extern "C" int64_t leclip_bpe_vocab_size(void* handle) { return handle ? (int64_t)((Bpe*)handle)->encoder.size() : LECLIP_E_INVALID; }

extern "C" int64_t leclip_bpe_encode(void* handle, const char* utf8, int64_t* ids, int64_t max_ids) {
    if (!handle || !utf8 || (!ids && max_ids > 0)) { leclip_set_error("bpe_encode: null argument"); return LECLIP_E_INVALID; }
    std::vector<int> v;
    const int rc = encode_text(*(Bpe*)handle, utf8, v);
    if (rc) return rc;
    for (int64_t i = 0; i < (int64_t)v.size() && i < max_ids; ++i) ids[i] = v[i];
    return (int64_t)v.size();
}

extern "C" int leclip_bpe_tokenize(void* handle, const char* const* texts, int64_t n, int context_length, int truncate, int64_t* out) {
    if (!handle || !texts || !out || n <= 0 || context_length < 2) { leclip_set_error("bpe_tokenize: bad argument"); return LECLIP_E_INVALID; }
    Bpe& b = *(Bpe*)handle;
    for (int64_t i = 0; i < n; ++i) {
        std::vector<int> v;
        const int rc = encode_text(b, texts[i], v);
        if (rc) return rc;
        int64_t* row = out + i * context_length;
        for (int k = 0; k < context_length; ++k) row[k] = 0;
        const int64_t total = (int64_t)v.size() + 2;
        if (total > context_length && !truncate) {
            leclip_set_error("Input %lld is too long for context length %d", (long long)i, context_length);     // clip.py:217-218
            return LECLIP_E_INVALID;
        }
        row[0] = b.sot;
        const int64_t keep = total > context_length ? context_length - 2 : (int64_t)v.size();
        for (int64_t k = 0; k < keep; ++k) row[1 + k] = v[k];
        row[total > context_length ? context_length - 1 : 1 + keep] = b.eot;
    }
    return LECLIP_OK;
}
