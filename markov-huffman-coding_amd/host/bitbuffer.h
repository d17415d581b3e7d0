// bitbuffer.h — bit-granular FILE* wrapper with the interface of the reference's bitbuffer
// (src/bitbuffer.h:19-64): same method names and semantics (MSB-first, write side zero-pads on flush,
// owns and closes the FILE* except stdout), so code written against the reference's coding.h face
// compiles against this one.  Only table files go through it here; payload bits are produced and
// consumed by the HIP kernels.
#ifndef MHC_HOST_BITBUFFER_H
#define MHC_HOST_BITBUFFER_H

#include <stdio.h>

#include <vector>

class bitbuffer {
public:
    enum e_mode { read, write };

    bitbuffer(FILE* file, e_mode mode) : file_(file), mode_(mode) {}
    bitbuffer(const bitbuffer&) = delete;
    bitbuffer& operator=(const bitbuffer&) = delete;
    ~bitbuffer() {
        if (mode_ == write) flush();
        if (file_ != stdout) fclose(file_);      // src/bitbuffer.h:35-40
    }

    // ---- write side (src/bitbuffer.cpp:9-43)
    void push_bit(int b) {
        cur_ = (unsigned char)(cur_ | ((b & 1) << (7 - nbit_)));
        if (++nbit_ == 8) emit();
    }
    void push_byte(unsigned char b) {
        for (int i = 7; i >= 0; --i) push_bit((b >> i) & 1);
    }
    void push_bytes(const unsigned char* p, size_t n) {   // fast path when byte aligned
        if (nbit_ == 0) { drain(); fwrite_checked(p, n); }
        else for (size_t i = 0; i < n; ++i) push_byte(p[i]);
    }
    int get_bi() const { return nbit_; }
    // rounds up to a whole byte, zero padded (src/bitbuffer.cpp:170-180)
    void flush() {
        if (nbit_) emit();
        drain();
    }

    // ---- read side (src/bitbuffer.cpp:75-140)
    unsigned char peek_bit() {
        fill();
        return rpos_ < rbuf_.size() ? (unsigned char)((rbuf_[rpos_] >> (7 - nbit_)) & 1) : 0;
    }
    unsigned char pop_bit() {
        unsigned char b = peek_bit();
        if (++nbit_ == 8) { nbit_ = 0; ++rpos_; }
        return b;
    }
    unsigned char pop_byte() {
        unsigned char v = 0;
        for (int i = 0; i < 8; ++i) v = (unsigned char)((v << 1) | pop_bit());
        return v;
    }
    // Everything that has not been consumed yet, starting at the current BYTE (the cursor must be byte
    // aligned or at bit 0 of a fresh file): used to hand a whole table file to the C ABI.
    std::vector<unsigned char> rest() {
        fill();
        return std::vector<unsigned char>(rbuf_.begin() + (long)rpos_, rbuf_.end());
    }

private:
    void emit() { wbuf_.push_back(cur_); cur_ = 0; nbit_ = 0; if (wbuf_.size() >= 32768) drain(); }
    void drain() { if (!wbuf_.empty()) { fwrite_checked(wbuf_.data(), wbuf_.size()); wbuf_.clear(); } }
    void fwrite_checked(const unsigned char* p, size_t n);
    void fill();

    FILE* file_;
    e_mode mode_;
    unsigned char cur_ = 0;
    int nbit_ = 0;
    std::vector<unsigned char> wbuf_;
    std::vector<unsigned char> rbuf_;
    size_t rpos_ = 0;
    bool loaded_ = false;
};

#endif
