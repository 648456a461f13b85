#include "png_io.h"

#include <zlib.h>

#include <cstring>
#include <fstream>
#include <istream>
#include <ostream>
#include <stdexcept>

namespace rta {
namespace {

const uint8_t kSignature[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};

// ---------------------------------------------------------------- chunk-level helpers

void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back(uint8_t(x >> 24));
    v.push_back(uint8_t(x >> 16));
    v.push_back(uint8_t(x >> 8));
    v.push_back(uint8_t(x));
}
void put_be16(std::vector<uint8_t>& v, uint16_t x) {
    v.push_back(uint8_t(x >> 8));
    v.push_back(uint8_t(x));
}

// Emits length | type | payload | crc(type+payload)
void emit_chunk(std::ostream& out, const char type[4], const std::vector<uint8_t>& payload) {
    std::vector<uint8_t> body(type, type + 4);
    body.insert(body.end(), payload.begin(), payload.end());
    std::vector<uint8_t> head;
    put_be32(head, uint32_t(payload.size()));
    const uint32_t crc = uint32_t(crc32(0L, body.data(), uInt(body.size())));
    std::vector<uint8_t> tail;
    put_be32(tail, crc);
    out.write(reinterpret_cast<const char*>(head.data()), std::streamsize(head.size()));
    out.write(reinterpret_cast<const char*>(body.data()), std::streamsize(body.size()));
    out.write(reinterpret_cast<const char*>(tail.data()), std::streamsize(tail.size()));
}

uint32_t get_be32(const uint8_t* p) {
    return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | uint32_t(p[3]);
}

struct Chunk {
    char type[5] = {0, 0, 0, 0, 0};
    std::vector<uint8_t> data;
};

bool read_exact(std::istream& in, void* dst, size_t n) {
    in.read(static_cast<char*>(dst), std::streamsize(n));
    return size_t(in.gcount()) == n;
}

Chunk read_chunk(std::istream& in) {
    uint8_t lenb[4];
    if (!read_exact(in, lenb, 4)) throw std::runtime_error("PNG file ended unexpectedly");
    const uint32_t len = get_be32(lenb);
    Chunk c;
    std::vector<uint8_t> body(size_t(len) + 4);
    if (!read_exact(in, body.data(), body.size())) throw std::runtime_error("PNG file ended unexpectedly");
    uint8_t crcb[4];
    if (!read_exact(in, crcb, 4)) throw std::runtime_error("PNG file ended unexpectedly");
    if (uint32_t(crc32(0L, body.data(), uInt(body.size()))) != get_be32(crcb)) {
        throw std::runtime_error("PNG chunk checksum wrong");
    }
    std::memcpy(c.type, body.data(), 4);
    c.data.assign(body.begin() + 4, body.end());
    return c;
}

struct Header {
    uint32_t width = 0, height = 0;
    bool alpha = false;
};

Header parse_ihdr(const Chunk& c) {
    if (c.data.size() != 13) throw std::runtime_error("unexpected PNG IHDR chunk length");
    Header h;
    h.width = get_be32(&c.data[0]);
    h.height = get_be32(&c.data[4]);
    const uint8_t depth = c.data[8], colour = c.data[9], comp = c.data[10], filt = c.data[11],
                  lace = c.data[12];
    if (h.width == 0 || h.height == 0) throw std::runtime_error("PNG has zero size");
    if (comp != 0) throw std::runtime_error("unexpected PNG compression method");
    if (filt != 0) throw std::runtime_error("unexpected PNG filter method");
    if (depth != 8) throw std::runtime_error("only PNG with bit depth 8 bits/color supported");
    if (colour != 2 && colour != 6) {
        throw std::runtime_error("only PNG with truecolor (and optionally alpha) supported");
    }
    if (lace != 0) throw std::runtime_error("only non-interlaced PNGs supported");
    h.alpha = colour == 6;
    return h;
}

int paeth_predict(int a, int b, int c) {
    const int p = a + b - c;
    const int pa = p > a ? p - a : a - p;
    const int pb = p > b ? p - b : b - p;
    const int pc = p > c ? p - c : c - p;
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

// Inflates one zlib stream of filtered scanlines and undoes the PNG filters.
ImageRGBA8 unfilter(const Header& h, const std::vector<uint8_t>& zdata) {
    const size_t bpp = h.alpha ? 4 : 3;
    const size_t stride = size_t(h.width) * bpp;
    std::vector<uint8_t> raw((stride + 1) * h.height);
    uLongf raw_len = uLongf(raw.size());
    const int ret = uncompress(raw.data(), &raw_len, zdata.data(), uLong(zdata.size()));
    if (ret == Z_MEM_ERROR) throw std::runtime_error("png uncompress: out of memory");
    if (ret == Z_BUF_ERROR) throw std::runtime_error("png uncompress: not enough memory reservered");
    if (ret == Z_DATA_ERROR) throw std::runtime_error("png uncompress: data corrupted");
    if (ret != Z_OK) throw std::runtime_error("png uncompress: unknown error");
    if (raw_len != raw.size()) throw std::runtime_error("png uncompress: data corrupted");

    ImageRGBA8 img;
    img.width = h.width;
    img.height = h.height;
    img.pixels.resize(size_t(h.width) * h.height * 4);
    std::vector<uint8_t> prev(stride, 0), cur(stride, 0);
    for (uint32_t y = 0; y < h.height; y++) {
        const uint8_t* line = &raw[(stride + 1) * y];
        const uint8_t ftype = line[0];
        if (ftype > 4) throw std::runtime_error("unsupported PNG filter type");
        for (size_t i = 0; i < stride; i++) {
            const int left = i >= bpp ? cur[i - bpp] : 0;
            const int up = prev[i];
            const int upleft = i >= bpp ? prev[i - bpp] : 0;
            int pred = 0;
            switch (ftype) {
                case 1: pred = left; break;
                case 2: pred = up; break;
                case 3: pred = (left + up) / 2; break;
                case 4: pred = paeth_predict(left, up, upleft); break;
                default: break;
            }
            cur[i] = uint8_t(line[1 + i] + pred);
        }
        uint8_t* dst = &img.pixels[size_t(y) * h.width * 4];
        for (uint32_t x = 0; x < h.width; x++) {
            dst[x * 4 + 0] = cur[x * bpp + 0];
            dst[x * 4 + 1] = cur[x * bpp + 1];
            dst[x * 4 + 2] = cur[x * bpp + 2];
            dst[x * 4 + 3] = h.alpha ? cur[x * bpp + 3] : 255;
        }
        prev.swap(cur);
    }
    return img;
}

void check_signature(std::istream& in) {
    uint8_t sig[8];
    if (!read_exact(in, sig, 8) || std::memcmp(sig, kSignature, 8) != 0) {
        throw std::runtime_error("wrong PNG header");
    }
}

// Sub-filters and deflates one frame (compress() = default level, as the reference does).
std::vector<uint8_t> deflate_frame(const uint8_t* rgba8, uint32_t width, uint32_t height) {
    const size_t stride = size_t(width) * 4;
    std::vector<uint8_t> filtered;
    filtered.reserve((stride + 1) * height);
    for (uint32_t y = 0; y < height; y++) {
        const uint8_t* row = rgba8 + stride * y;
        filtered.push_back(1);   // Sub
        for (size_t i = 0; i < stride; i++) {
            const uint8_t left = i >= 4 ? row[i - 4] : 0;
            filtered.push_back(uint8_t(row[i] - left));
        }
    }
    uLongf zlen = compressBound(uLong(filtered.size()));
    std::vector<uint8_t> z(zlen);
    if (compress(z.data(), &zlen, filtered.data(), uLong(filtered.size())) != Z_OK) {
        throw std::runtime_error("zlib compression failed");
    }
    z.resize(zlen);
    return z;
}

void emit_preamble(std::ostream& out, uint32_t width, uint32_t height) {
    out.write(reinterpret_cast<const char*>(kSignature), 8);
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, width);
    put_be32(ihdr, height);
    const uint8_t tail[5] = {8, 6, 0, 0, 0};   // 8 bit, RGBA, deflate, filter method 0, no interlace
    ihdr.insert(ihdr.end(), tail, tail + 5);
    emit_chunk(out, "IHDR", ihdr);
    std::vector<uint8_t> gama;
    put_be32(gama, 45455u);   // 1/2.2
    emit_chunk(out, "gAMA", gama);
    std::vector<uint8_t> chrm;   // sRGB white point and primaries
    for (uint32_t v : {31270u, 32900u, 64000u, 33000u, 30000u, 60000u, 15000u, 6000u}) put_be32(chrm, v);
    emit_chunk(out, "cHRM", chrm);
}

void emit_trailer(std::ostream& out) {
    // zTXt "Software" = the same text the reference stamps, deflated at level 9
    static const char kText[] = "clemy.org";
    std::vector<uint8_t> ztxt;
    for (const char* p = "Software"; *p; p++) ztxt.push_back(uint8_t(*p));
    ztxt.push_back(0);   // keyword terminator
    ztxt.push_back(0);   // compression method 0
    uLongf zlen = compressBound(sizeof kText - 1);
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, reinterpret_cast<const Bytef*>(kText), sizeof kText - 1, 9) != Z_OK) {
        throw std::runtime_error("zlib compression failed");
    }
    ztxt.insert(ztxt.end(), z.begin(), z.begin() + zlen);
    emit_chunk(out, "zTXt", ztxt);
    emit_chunk(out, "IEND", {});
}

}  // namespace

// ---------------------------------------------------------------- reading

ImageRGBA8 read_png(std::istream& in) {
    check_signature(in);
    bool have_header = false;
    Header h;
    std::vector<uint8_t> zdata;
    for (;;) {
        const Chunk c = read_chunk(in);
        if (std::strcmp(c.type, "IHDR") == 0) {
            if (have_header) throw std::runtime_error("unexpected PNG IHDR chunk");
            h = parse_ihdr(c);
            have_header = true;
        } else if (std::strcmp(c.type, "IDAT") == 0) {
            if (!have_header) throw std::runtime_error("unexpected PNG IDAT chunk");
            zdata.insert(zdata.end(), c.data.begin(), c.data.end());
        } else if (std::strcmp(c.type, "IEND") == 0) {
            if (!have_header) throw std::runtime_error("unexpected PNG IEND chunk");
            if (!c.data.empty()) throw std::runtime_error("unexpected PNG IEND chunk length");
            break;
        }
        // every other chunk is skipped (its CRC was still checked)
    }
    return unfilter(h, zdata);
}

ImageRGBA8 read_png_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("texture file \"" + path + "\" could not be opened");
    return read_png(f);
}

std::vector<ImageRGBA8> read_apng_frames(std::istream& in) {
    check_signature(in);
    bool have_header = false;
    Header h;
    std::vector<ImageRGBA8> frames;
    std::vector<uint8_t> zdata;
    auto flush = [&]() {
        if (!zdata.empty()) {
            frames.push_back(unfilter(h, zdata));
            zdata.clear();
        }
    };
    for (;;) {
        const Chunk c = read_chunk(in);
        if (std::strcmp(c.type, "IHDR") == 0) {
            h = parse_ihdr(c);
            have_header = true;
        } else if (std::strcmp(c.type, "fcTL") == 0) {
            flush();
        } else if (std::strcmp(c.type, "IDAT") == 0) {
            if (!have_header) throw std::runtime_error("unexpected PNG IDAT chunk");
            zdata.insert(zdata.end(), c.data.begin(), c.data.end());
        } else if (std::strcmp(c.type, "fdAT") == 0) {
            if (!have_header || c.data.size() < 4) throw std::runtime_error("unexpected PNG fdAT chunk");
            zdata.insert(zdata.end(), c.data.begin() + 4, c.data.end());
        } else if (std::strcmp(c.type, "IEND") == 0) {
            flush();
            break;
        }
    }
    return frames;
}

// ---------------------------------------------------------------- writing

void quantize_rgba(const float* rgba, uint64_t n_pixels, float gain, uint8_t* out) {
    for (uint64_t i = 0; i < n_pixels * 4; i++) out[i] = quantize_channel(rgba[i], gain);
}

void write_png(std::ostream& out, const uint8_t* rgba8, uint32_t width, uint32_t height) {
    emit_preamble(out, width, height);
    emit_chunk(out, "IDAT", deflate_frame(rgba8, width, height));
    emit_trailer(out);
}

ApngWriter::ApngWriter(std::ostream& out, uint32_t width, uint32_t height, uint32_t frame_count)
    : out_(out), width_(width), height_(height) {
    emit_preamble(out_, width, height);
    std::vector<uint8_t> actl;
    put_be32(actl, frame_count);
    put_be32(actl, 0);   // loop forever
    emit_chunk(out_, "acTL", actl);
}

std::vector<uint8_t> deflate_rgba8(const uint8_t* rgba8, uint32_t width, uint32_t height) { return deflate_frame(rgba8, width, height); }

void ApngWriter::frame(const uint8_t* rgba8, uint32_t frame_index, float fps) { frame_deflated(deflate_frame(rgba8, width_, height_), frame_index, fps); }

void ApngWriter::frame_deflated(const std::vector<uint8_t>& z, uint32_t frame_index, float fps) {
    // sequence numbers: fcTL of frame f is 0 for f == 0 else 2f-1; fdAT of frame f>0 is 2f
    std::vector<uint8_t> fctl;
    put_be32(fctl, frame_index == 0 ? 0 : frame_index * 2 - 1);
    put_be32(fctl, width_);
    put_be32(fctl, height_);
    put_be32(fctl, 0);   // x offset
    put_be32(fctl, 0);   // y offset
    put_be16(fctl, static_cast<uint16_t>(1000.0f / fps));
    put_be16(fctl, 1000);
    fctl.push_back(0);   // dispose: none
    fctl.push_back(0);   // blend: source
    emit_chunk(out_, "fcTL", fctl);
    if (frame_index == 0) {
        emit_chunk(out_, "IDAT", z);
    } else {
        std::vector<uint8_t> payload;
        put_be32(payload, frame_index * 2);
        payload.insert(payload.end(), z.begin(), z.end());
        emit_chunk(out_, "fdAT", payload);
    }
}

void ApngWriter::end() { emit_trailer(out_); }

}  // namespace rta
