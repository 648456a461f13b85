// PNG texture reader and PNG/APNG frame writer (zlib-backed), host I/O on the reference's side of
// the boundary.  Byte formats follow reference src/pngloader.cpp (reader subset) and
// reference src/pngwriter.cpp:171-241 (writer: RGBA8, Sub filter, gAMA+cHRM, one IDAT/fdAT per
// frame, zTXt Software footer) so that files are interchangeable with the reference's.
#pragma once

#include <cstdint>
#include <iosfwd>
#include <string>
#include <vector>

namespace rta {

struct ImageRGBA8 {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> pixels;   // width*height*4, rows top-down
};

// 8-bit truecolour (RGB or RGBA), non-interlaced only; every chunk CRC is verified.
ImageRGBA8 read_png(std::istream& in);
ImageRGBA8 read_png_file(const std::string& path);

// All frames of an APNG (or the single image of a PNG) written by write_png/ApngWriter below or by
// the reference; used by tests to compare renders.  Only full-size frames with Sub/None filters
// written as one IDAT/fdAT each are guaranteed to decode.
std::vector<ImageRGBA8> read_apng_frames(std::istream& in);

// Quantiser of the reference (Color::scaleOut, src/types.h:274-282): u8(clamp(v*gain,0,1)*255.0f).
inline uint8_t quantize_channel(float v, float gain) {
    float s = v * gain;
    s = s < 0.0f ? 0.0f : (s > 1.0f ? 1.0f : s);   // std::clamp (NaN stays NaN -> 0 on x86 cast)
    return static_cast<uint8_t>(s * 255.0f);
}
void quantize_rgba(const float* rgba, uint64_t n_pixels, float gain, uint8_t* out);

// Still image.
void write_png(std::ostream& out, const uint8_t* rgba8, uint32_t width, uint32_t height);

// The compressed scanlines of one frame (Sub filter per row, one zlib compress() at its default level): the costly
// part of writing a frame, a pure function of the pixels, so frames can be compressed on worker threads while the
// next ones render and still be appended in order.
std::vector<uint8_t> deflate_rgba8(const uint8_t* rgba8, uint32_t width, uint32_t height);

// Animation: begin(), then frame() (or frame_deflated() with deflate_rgba8's result) once per frame in order, then end().
class ApngWriter {
public:
    ApngWriter(std::ostream& out, uint32_t width, uint32_t height, uint32_t frame_count);
    void frame(const uint8_t* rgba8, uint32_t frame_index, float fps);
    void frame_deflated(const std::vector<uint8_t>& z, uint32_t frame_index, float fps);
    void end();

private:
    std::ostream& out_;
    uint32_t width_, height_;
};

}  // namespace rta
