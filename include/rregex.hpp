// rregex.hpp — the reference's C++ interface for the hot path, rebuilt on the C ABI (rrx.h).
//
// Same names, argument meaning and error behaviour as src/inc/regex.h of the reference, so that a caller
// written against it (src/test/main.cpp:22-31) compiles unchanged:
//
//     Regex::RRegex r(pattern);                              // regex.h:212-228, throws std::runtime_error
//     auto it = r.get_acceptance_iter(text)++;               // regex.h:225-227, 118, 156-159
//     bool is_match = (*it).has_value();                     // regex.h:119, 160-162
//     std::string s = (*it)->str();                          // regex.h:100-105
//
// plus the batch entry the reference lacks (RRegex::match_lines / match_corpus).  Header-only; link librrx.so.
// There is no CPU matcher behind it: matching runs on the gfx950 device given to the RRegex constructor.
#pragma once
#include <cstring>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "rrx.h"

namespace Regex {

class Match {                                        // regex.h:100-105
public:
    char *start;
    char *end;
    std::string str() { return std::string(start, end - start); }
};

class RegexIterator {                                // regex.h:106-112 (+ the virtual destructor it lacks)
public:
    virtual ~RegexIterator() = default;
    virtual RegexIterator &operator++(int) = 0;
    virtual RegexIterator *create_copy() = 0;
    virtual std::optional<Match> operator*() = 0;
    virtual void print() = 0;
};

class IteratorWrapper : public RegexIterator {       // regex.h:113-122
    std::unique_ptr<RegexIterator> concrete;
public:
    IteratorWrapper(RegexIterator *ptr) : concrete(ptr) {}
    IteratorWrapper(IteratorWrapper &to_copy) : concrete(to_copy.create_copy()) {}
    IteratorWrapper(IteratorWrapper &&) = default;
    IteratorWrapper &operator++(int) override { (*concrete)++; return *this; }
    std::optional<Match> operator*() override { return **concrete; }
    void print() override { concrete->print(); }
    RegexIterator *create_copy() override { return concrete->create_copy(); }
};

class IterFactoryBase {                              // regex.h:123-126: the plugin seam
public:
    virtual ~IterFactoryBase() = default;
    virtual IteratorWrapper get_acceptance_iter(char *c) = 0;
};

// The MI355X engine as one more IterFactoryBase-shaped implementation (beside NFA<S>::IterFactory, regex.h:168-175).
class DeviceIterFactory : public IterFactoryBase {
    rrx_regex *re_;
    int device_;

    static void check(int rc) { if (rc != RRX_OK) throw std::runtime_error(rrx_last_error()); }

    class AcceptanceIterator : public RegexIterator {        // regex.h:150-165
        const rrx_regex *re_;
        int device_;
        char *initial_;
        char *current_;
        bool consumed_ = false, accepted_ = false;
    public:
        AcceptanceIterator(const rrx_regex *re, int device, char *text) : re_(re), device_(device), initial_(text), current_(text) {}
        RegexIterator &operator++(int) override {            // consume up to the terminator; idempotent afterwards
            if (!consumed_) {
                int acc = 0;
                size_t len = 0;
                check(rrx_match_cstr(re_, device_, initial_, &acc, &len));
                accepted_ = acc != 0;
                current_ = initial_ + len;
                consumed_ = true;
            }
            return *this;
        }
        std::optional<Match> operator*() override {
            // before ++ the state set is {initial}: only a final initial state accepts (NFA.cc:103-107)
            const bool ok = consumed_ ? accepted_ : rrx_accepts_empty(re_) != 0;
            return ok ? std::optional<Match>(Match{initial_, current_}) : std::nullopt;
        }
        RegexIterator *create_copy() override { return new AcceptanceIterator(*this); }
        void print() override {}                             // the reference dumps its tables here (NFA.cc:14-41)
    };

public:
    DeviceIterFactory(const char *pattern, int device, int engine = RRX_ENGINE_AUTO) : re_(nullptr), device_(device) {
        check(rrx_compile_ex(pattern, engine, &re_));
    }
    ~DeviceIterFactory() override { rrx_free(re_); }
    DeviceIterFactory(const DeviceIterFactory &) = delete;
    DeviceIterFactory &operator=(const DeviceIterFactory &) = delete;
    IteratorWrapper get_acceptance_iter(char *c) override { return IteratorWrapper(new AcceptanceIterator(re_, device_, c)); }
    const rrx_regex *handle() const { return re_; }
    int device() const { return device_; }
};

class RRegex {                                       // regex.h:212-228
public:
    std::unique_ptr<IterFactoryBase> iter_factory;
    RRegex(const char *p, int device = 0) : iter_factory(std::make_unique<DeviceIterFactory>(p, device)) {}
    IteratorWrapper get_acceptance_iter(char *c) { return iter_factory->get_acceptance_iter(c); }

    // ---- batch entries (not in the reference) ------------------------------------------------------
    // accept[i] for every '\n'-delimited string of a HOST buffer (upload + index + match + download).
    std::vector<uint8_t> match_lines(const char *bytes, size_t nbytes) {
        auto *f = static_cast<DeviceIterFactory *>(iter_factory.get());
        size_t cap = 1;
        for (size_t i = 0; i < nbytes; i++) cap += bytes[i] == '\n';
        std::vector<uint8_t> out(cap);
        size_t n = 0;
        if (rrx_match_host(f->handle(), f->device(), bytes, nbytes, out.data(), cap, &n) != RRX_OK)
            throw std::runtime_error(rrx_last_error());
        out.resize(n);
        return out;
    }
    // the hot path proper: device-resident corpus in, accept bitmap out, asynchronous on `stream`
    void match_corpus(const rrx_corpus *corpus, uint32_t *d_accept_bits, void *stream = nullptr) {
        auto *f = static_cast<DeviceIterFactory *>(iter_factory.get());
        if (rrx_match_corpus(f->handle(), corpus, d_accept_bits, stream) != RRX_OK) throw std::runtime_error(rrx_last_error());
    }
    // the same for a device buffer that has no rrx_corpus yet: ONE pass over the text (no index pass); returns the number
    // of strings; d_accept_bits holds cap_words words.  Synchronous.
    size_t match_device(const void *d_bytes, size_t nbytes, uint32_t *d_accept_bits, size_t cap_words, void *stream = nullptr) {
        auto *f = static_cast<DeviceIterFactory *>(iter_factory.get());
        size_t nlines = 0;
        if (rrx_match_device(f->handle(), f->device(), d_bytes, nbytes, d_accept_bits, cap_words, &nlines, stream) != RRX_OK)
            throw std::runtime_error(rrx_last_error());
        return nlines;
    }
    // search: per string the accepted substring [d_start[i], d_end[i]) with the smallest end, then the smallest start
    // (offsets relative to the string; 0xFFFFFFFF = none).  The reference's README promises this, its code has not got it.
    void search_corpus(const rrx_corpus *corpus, uint32_t *d_start, uint32_t *d_end, void *stream = nullptr) {
        auto *f = static_cast<DeviceIterFactory *>(iter_factory.get());
        if (rrx_search_corpus(f->handle(), corpus, d_start, d_end, stream) != RRX_OK) throw std::runtime_error(rrx_last_error());
    }
    // a batch of explicit items (an offsets array over one buffer) indexed once with rrx_items_create: one byte per item
    void match_items(const rrx_items *items, uint8_t *d_accept, void *stream = nullptr) {
        auto *f = static_cast<DeviceIterFactory *>(iter_factory.get());
        if (rrx_match_items(f->handle(), items, d_accept, stream) != RRX_OK) throw std::runtime_error(rrx_last_error());
    }
    // every lazy match of every string, left to right, in one call (rrx_search_all): d_first[i] (nlines + 1 entries) = slot
    // of string i's first match; returns the number of matches - if it exceeds `cap` (the entries d_start / d_end hold),
    // those beyond were not written: call again with arrays of that size.  Synchronous.
    size_t search_all(const rrx_corpus *corpus, uint64_t *d_first, uint32_t *d_start, uint32_t *d_end, size_t cap, void *stream = nullptr) {
        auto *f = static_cast<DeviceIterFactory *>(iter_factory.get());
        size_t total = 0;
        if (rrx_search_all(f->handle(), corpus, d_first, d_start, d_end, cap, &total, stream) != RRX_OK) throw std::runtime_error(rrx_last_error());
        return total;
    }
};

}  // namespace Regex
