// Baseline JPEG (ITU-T T.81, SOF0 / SOF1 Huffman, 8 bit) -> RGB8: what Texture::new_image (texture.rs:25-31:
// ImageReader::open(path)?.decode()?.to_rgb8()) needs for the reference's one image asset (assets/earthmap.jpg: SOF0,
// 1024 x 512, three components at 1x1 sampling, no restart interval).
//
// The reference decodes through the `image` crate 0.23.13 -> `jpeg-decoder` 0.1.22 (Cargo.lock); neither is in
// /root/reference, so this is a restatement of the published algorithm, not of their source: T.81 Annex F Huffman
// decoding, dequantisation, the Loeffler-Ligtenberg-Moschytz 8x8 inverse DCT in 12-bit fixed point (the public-domain
// "islow" form that crate also uses), JFIF YCbCr -> RGB (ITU-R BT.601 full range).  Texels may differ from the crate's by
// +-1 LSB (IDCT and colour rounding): parity unpinned against it; pinned here against an independent decode of the same
// file (tests/test_jpeg.py).  Subsampled chroma (2x1, 1x2, 2x2) is upsampled by replication (the crate interpolates):
// supported for convenience, not claimed to match.  Progressive, arithmetic-coded, 12-bit and CMYK files are refused.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace rth {
namespace jpeg {

struct Huff {
  // canonical code tables of T.81 Annex C / F.2.2.3: codes of length l are mincode[l] .. maxcode[l]
  int32_t mincode[17], maxcode[18], valptr[17];
  uint8_t vals[256];
  bool present = false;
};

struct Component {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int pred = 0;
  int stride = 0, rows = 0;  // of the component plane (whole MCUs)
  std::vector<uint8_t> plane;
};

class Decoder {
 public:
  Decoder(const uint8_t* data, size_t size) : d_(data), n_(size) {}

  // returns RGB8 row-major; throws std::runtime_error with a reason otherwise
  std::vector<uint8_t> decode(uint32_t& width, uint32_t& height) {
    if (n_ < 4 || d_[0] != 0xFF || d_[1] != 0xD8) fail("not a JPEG file (no SOI marker)");
    pos_ = 2;
    bool have_frame = false, done = false;
    while (!done) {
      const int m = next_marker();
      switch (m) {
        case 0xC0:
        case 0xC1:
          if (have_frame) fail("a second SOF segment (hierarchical / multi-frame JPEG is not supported)");
          read_frame();
          have_frame = true;
          break;
        case 0xC2:
          fail("progressive JPEG (SOF2) is not supported: re-save the image as baseline");
        case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
          fail("lossless / hierarchical / arithmetic-coded JPEG is not supported");
        case 0xC4:
          read_dht();
          break;
        case 0xDB:
          read_dqt();
          break;
        case 0xDD: {
          const size_t len = segment_length();
          if (len != 4) fail("bad DRI segment");
          restart_interval_ = (d_[pos_ + 2] << 8) | d_[pos_ + 3];
          pos_ += len;
          break;
        }
        case 0xDA:
          if (!have_frame) fail("SOS before SOF");
          read_scan();
          done = true;  // baseline, one interleaved scan (what every encoder writes); a second scan is refused below
          break;
        case 0xD9:
          fail("end of image before any scan");
        default:  // APPn, COM and anything else with a length
          pos_ += segment_length();
      }
    }
    width = (uint32_t)w_;
    height = (uint32_t)h_;
    return to_rgb();
  }

 private:
  const uint8_t* d_;
  size_t n_, pos_ = 0;
  int w_ = 0, h_ = 0, hmax_ = 1, vmax_ = 1, restart_interval_ = 0;
  std::vector<Component> comp_;
  uint16_t q_[4][64];
  bool q_present_[4] = {false, false, false, false};
  Huff dc_[4], ac_[4];
  // entropy-coded segment reader
  uint32_t bits_ = 0;
  int nbits_ = 0;
  bool hit_marker_ = false;

  [[noreturn]] static void fail(const std::string& why) { throw std::runtime_error(why); }

  int next_marker() {
    while (pos_ < n_ && d_[pos_] != 0xFF) ++pos_;
    while (pos_ < n_ && d_[pos_] == 0xFF) ++pos_;
    if (pos_ >= n_) fail("truncated file (marker expected)");
    return d_[pos_++];
  }
  size_t segment_length() {
    if (pos_ + 2 > n_) fail("truncated segment");
    const size_t len = ((size_t)d_[pos_] << 8) | d_[pos_ + 1];
    if (len < 2 || pos_ + len > n_) fail("truncated segment");
    return len;
  }

  void read_dqt() {
    const size_t end = pos_ + segment_length();
    size_t p = pos_ + 2;
    while (p < end) {
      const int pq = d_[p] >> 4, tq = d_[p] & 15;
      ++p;
      if (tq > 3 || pq > 1 || p + (pq ? 128 : 64) > end) fail("bad DQT segment");
      if (pq == 1) fail("16-bit quantisation tables are not supported (8-bit baseline JPEG only)");
      for (int i = 0; i < 64; ++i) {  // stored in zig-zag order; kept that way (the coefficients arrive in it too)
        q_[tq][i] = pq ? (uint16_t)((d_[p] << 8) | d_[p + 1]) : d_[p];
        p += pq ? 2 : 1;
      }
      q_present_[tq] = true;
    }
    pos_ = end;
  }

  void read_dht() {
    const size_t end = pos_ + segment_length();
    size_t p = pos_ + 2;
    while (p < end) {
      const int tc = d_[p] >> 4, th = d_[p] & 15;
      ++p;
      if (tc > 1 || th > 3 || p + 16 > end) fail("bad DHT segment");
      Huff& t = tc ? ac_[th] : dc_[th];
      int counts[17] = {0};
      int total = 0;
      for (int l = 1; l <= 16; ++l) total += counts[l] = d_[p + l - 1];
      p += 16;
      if (total > 256 || p + total > end) fail("bad DHT segment");
      std::memcpy(t.vals, d_ + p, (size_t)total);
      p += total;
      int code = 0, k = 0;
      for (int l = 1; l <= 16; ++l) {  // T.81 Figure F.15 (Decoder_tables)
        t.valptr[l] = k;
        t.mincode[l] = code;
        code += counts[l];
        k += counts[l];
        t.maxcode[l] = counts[l] ? code - 1 : -1;
        code <<= 1;
      }
      t.maxcode[17] = 0x7FFFFFFF;
      t.present = true;
    }
    pos_ = end;
  }

  void read_frame() {
    const size_t len = segment_length();
    if (len < 8) fail("bad SOF segment");
    const uint8_t* s = d_ + pos_ + 2;
    if (s[0] != 8) fail("only 8-bit JPEG is supported");
    h_ = (s[1] << 8) | s[2];
    w_ = (s[3] << 8) | s[4];
    const int nc = s[5];
    if (w_ <= 0 || h_ <= 0) fail("empty image");
    if (nc != 1 && nc != 3) fail("only grey and YCbCr JPEG is supported (" + std::to_string(nc) + " components)");
    if (len != (size_t)(8 + 3 * nc)) fail("bad SOF segment");
    comp_.assign(nc, Component());
    hmax_ = vmax_ = 1;
    for (int i = 0; i < nc; ++i) {
      Component& c = comp_[i];
      c.id = s[6 + 3 * i];
      c.h = s[7 + 3 * i] >> 4;
      c.v = s[7 + 3 * i] & 15;
      c.tq = s[8 + 3 * i];
      if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) fail("unsupported sampling factors");
      hmax_ = c.h > hmax_ ? c.h : hmax_;
      vmax_ = c.v > vmax_ ? c.v : vmax_;
    }
    if (nc == 1) comp_[0].h = comp_[0].v = hmax_ = vmax_ = 1;  // (a single component is never interleaved)
    for (Component& c : comp_)
      if (hmax_ % c.h || vmax_ % c.v) fail("unsupported sampling factors");
    pos_ += len;
  }

  // ---- entropy-coded data (T.81 F.2.2) ----
  void fill() {
    while (nbits_ <= 24) {
      uint32_t byte = 0;
      if (!hit_marker_ && pos_ < n_) {
        byte = d_[pos_];
        if (byte == 0xFF) {
          const uint8_t nx = pos_ + 1 < n_ ? d_[pos_ + 1] : 0xD9;
          if (nx == 0x00) {
            pos_ += 2;  // stuffed zero
          } else {
            hit_marker_ = true;  // a marker: feed zeros from here on (F.2.2.5)
            byte = 0;
          }
        } else {
          ++pos_;
        }
      } else {
        hit_marker_ = true;
      }
      bits_ |= byte << (24 - nbits_);
      nbits_ += 8;
    }
  }
  int get_bits(int n) {
    if (n == 0) return 0;
    if (nbits_ < n) fill();
    const int v = (int)(bits_ >> (32 - n));
    bits_ <<= n;
    nbits_ -= n;
    return v;
  }
  int decode_symbol(const Huff& t) {  // T.81 Figure F.16
    if (!t.present) fail("scan uses a Huffman table the file does not define");
    int code = get_bits(1), l = 1;
    while (code > t.maxcode[l]) {
      if (++l > 16) fail("corrupt entropy-coded data (no such Huffman code)");
      code = (code << 1) | get_bits(1);
    }
    return t.vals[t.valptr[l] + code - t.mincode[l]];
  }
  static int extend(int v, int t) { return t == 0 ? 0 : (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v); }  // Figure F.12

  static uint8_t clamp8(int64_t x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

  // 8x8 inverse DCT, Loeffler-Ligtenberg-Moschytz, 12-bit fixed-point constants; columns first (to 10 fractional
  // bits dropped), then rows with the +128 level shift and the final rounding folded into one add.
  static void idct(const int32_t* in /* natural order, dequantised */, uint8_t* out, int out_stride) {
    constexpr auto fx = [](double x) { return (int32_t)(x * 4096.0 + 0.5); };
    constexpr int32_t c0_541 = fx(0.5411961), c1_847 = fx(1.847759065), c0_765 = fx(0.765366865), c1_175 = fx(1.175875602),
                      c0_298 = fx(0.298631336), c2_053 = fx(2.053119869), c3_072 = fx(3.072711026), c1_501 = fx(1.501321110),
                      c0_899 = fx(0.899976223), c2_562 = fx(2.562915447), c1_961 = fx(1.961570560), c0_390 = fx(0.390180644);
    int64_t tmp[64];  // (64-bit intermediates: a corrupt stream may carry coefficients far outside what an encoder
                      // produces, and the products below must not overflow — same values as 32-bit arithmetic on a valid file)
    auto pass = [&](int64_t s0, int64_t s1, int64_t s2, int64_t s3, int64_t s4, int64_t s5, int64_t s6, int64_t s7, int64_t bias,
                    int64_t* e /* x0 x1 x2 x3 */, int64_t* o /* t0 t1 t2 t3 */) {
      int64_t p1 = (s2 + s6) * c0_541;
      const int64_t t2 = p1 - s6 * c1_847, t3 = p1 + s2 * c0_765;
      const int64_t t0 = (s0 + s4) * 4096, t1 = (s0 - s4) * 4096;
      e[0] = t0 + t3 + bias;
      e[3] = t0 - t3 + bias;
      e[1] = t1 + t2 + bias;
      e[2] = t1 - t2 + bias;
      int64_t q0 = s7, q1 = s5, q2 = s3, q3 = s1;
      const int64_t p3 = q0 + q2, p4 = q1 + q3, p1b = q0 + q3, p2b = q1 + q2;
      const int64_t p5 = (p3 + p4) * c1_175;
      q0 *= c0_298;
      q1 *= c2_053;
      q2 *= c3_072;
      q3 *= c1_501;
      const int64_t r1 = p5 - p1b * c0_899, r2 = p5 - p2b * c2_562, r3 = -p3 * c1_961, r4 = -p4 * c0_390;
      o[3] = q3 + r1 + r4;
      o[2] = q2 + r2 + r3;
      o[1] = q1 + r2 + r4;
      o[0] = q0 + r1 + r3;
    };
    for (int c = 0; c < 8; ++c) {
      const int32_t* s = in + c;
      int64_t* t = tmp + c;
      if (!(s[8] | s[16] | s[24] | s[32] | s[40] | s[48] | s[56])) {  // a constant column
        const int64_t dc = (int64_t)s[0] * 4;
        for (int r = 0; r < 8; ++r) t[8 * r] = dc;
        continue;
      }
      int64_t e[4], o[4];
      pass(s[0], s[8], s[16], s[24], s[32], s[40], s[48], s[56], 512, e, o);
      t[0] = (e[0] + o[3]) >> 10;
      t[56] = (e[0] - o[3]) >> 10;
      t[8] = (e[1] + o[2]) >> 10;
      t[48] = (e[1] - o[2]) >> 10;
      t[16] = (e[2] + o[1]) >> 10;
      t[40] = (e[2] - o[1]) >> 10;
      t[24] = (e[3] + o[0]) >> 10;
      t[32] = (e[3] - o[0]) >> 10;
    }
    for (int r = 0; r < 8; ++r) {
      const int64_t* t = tmp + 8 * r;
      uint8_t* q = out + (size_t)r * out_stride;
      int64_t e[4], o[4];
      pass(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], 65536 + (128 << 17), e, o);
      q[0] = clamp8((e[0] + o[3]) >> 17);
      q[7] = clamp8((e[0] - o[3]) >> 17);
      q[1] = clamp8((e[1] + o[2]) >> 17);
      q[6] = clamp8((e[1] - o[2]) >> 17);
      q[2] = clamp8((e[2] + o[1]) >> 17);
      q[5] = clamp8((e[2] - o[1]) >> 17);
      q[3] = clamp8((e[3] + o[0]) >> 17);
      q[4] = clamp8((e[3] - o[0]) >> 17);
    }
  }

  void decode_block(Component& c, uint8_t* out) {
    static const uint8_t zigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                       41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                       30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    int32_t coef[64];
    std::memset(coef, 0, sizeof(coef));
    const uint16_t* q = q_[c.tq];
    const int t = decode_symbol(dc_[c.td]);
    if (t > 11) fail("corrupt entropy-coded data (DC category)");
    c.pred += extend(get_bits(t), t);
    // a conforming stream keeps the predictor inside 16 bits (T.81 F.1.2.1); a corrupt DC run must not be able to
    // overflow the int32 products below and in the inverse DCT
    if (c.pred < -32768 || c.pred > 32767) fail("corrupt entropy-coded data (DC predictor out of range)");
    coef[0] = c.pred * q[0];
    for (int k = 1; k < 64;) {  // T.81 Figure F.13
      const int rs = decode_symbol(ac_[c.ta]);
      const int r = rs >> 4, s = rs & 15;
      if (s == 0) {
        if (r != 15) break;  // end of block
        k += 16;
        continue;
      }
      k += r;
      if (k > 63) fail("corrupt entropy-coded data (run past the block)");
      coef[zigzag[k]] = extend(get_bits(s), s) * q[k];  // |.| <= 32767 * 255: fits; the inverse DCT works in 64 bits
      ++k;
    }
    idct(coef, out, c.stride);
  }

  void read_scan() {
    const size_t len = segment_length();
    if (len < 3) fail("bad SOS segment");  // (before its first byte is read: a length-2 SOS may end the file)
    const uint8_t* s = d_ + pos_ + 2;
    const int ns = s[0];
    if (ns != (int)comp_.size() || len != (size_t)(6 + 2 * ns))
      fail("only a single interleaved scan of all components is supported");
    for (int i = 0; i < ns; ++i) {
      Component* c = nullptr;
      for (Component& k : comp_)
        if (k.id == s[1 + 2 * i]) c = &k;
      if (!c) fail("scan names a component the frame does not have");
      c->td = s[2 + 2 * i] >> 4;
      c->ta = s[2 + 2 * i] & 15;
      if (c->td > 3 || c->ta > 3) fail("bad SOS segment");
      if (!q_present_[c->tq]) fail("frame uses a quantisation table the file does not define");
    }
    pos_ += len;
    const int mcu_w = 8 * hmax_, mcu_h = 8 * vmax_;
    const int mcus_x = (w_ + mcu_w - 1) / mcu_w, mcus_y = (h_ + mcu_h - 1) / mcu_h;
    for (Component& c : comp_) {
      c.stride = mcus_x * c.h * 8;
      c.rows = mcus_y * c.v * 8;
      c.plane.assign((size_t)c.stride * c.rows, 0);
      c.pred = 0;
    }
    bits_ = 0;
    nbits_ = 0;
    hit_marker_ = false;
    int until_restart = restart_interval_, next_rst = 0;
    for (int my = 0; my < mcus_y; ++my) {
      for (int mx = 0; mx < mcus_x; ++mx) {
        if (restart_interval_ && until_restart == 0) {
          // byte-align, expect RSTn (T.81 F.2.2.5 / E.2.4)
          bits_ = 0;
          nbits_ = 0;
          hit_marker_ = false;
          while (pos_ < n_ && d_[pos_] != 0xFF) ++pos_;
          while (pos_ < n_ && d_[pos_] == 0xFF) ++pos_;
          if (pos_ >= n_ || d_[pos_] != 0xD0 + next_rst) fail("restart marker missing");
          ++pos_;
          next_rst = (next_rst + 1) & 7;
          until_restart = restart_interval_;
          for (Component& c : comp_) c.pred = 0;
        }
        for (Component& c : comp_)
          for (int by = 0; by < c.v; ++by)
            for (int bx = 0; bx < c.h; ++bx)
              decode_block(c, c.plane.data() + (size_t)((my * c.v + by) * 8) * c.stride + (size_t)(mx * c.h + bx) * 8);
        if (restart_interval_) --until_restart;
      }
    }
  }

  std::vector<uint8_t> to_rgb() const {
    std::vector<uint8_t> rgb((size_t)w_ * h_ * 3);
    if (comp_.size() == 1) {
      const Component& y = comp_[0];
      for (int r = 0; r < h_; ++r)
        for (int x = 0; x < w_; ++x) {
          const uint8_t v = y.plane[(size_t)r * y.stride + x];
          uint8_t* o = &rgb[((size_t)r * w_ + x) * 3];
          o[0] = o[1] = o[2] = v;
        }
      return rgb;
    }
    const Component &cy = comp_[0], &cb = comp_[1], &cr = comp_[2];
    auto at = [&](const Component& c, int r, int x) {  // replication where a component is subsampled
      return (int)c.plane[(size_t)(r * c.v / vmax_) * c.stride + (size_t)(x * c.h / hmax_)];
    };
    for (int r = 0; r < h_; ++r)
      for (int x = 0; x < w_; ++x) {
        // JFIF 1.02: R = Y + 1.402 (Cr-128), G = Y - 0.34414 (Cb-128) - 0.71414 (Cr-128), B = Y + 1.772 (Cb-128)
        const float yy = (float)at(cy, r, x), u = (float)at(cb, r, x) - 128.0f, v = (float)at(cr, r, x) - 128.0f;
        uint8_t* o = &rgb[((size_t)r * w_ + x) * 3];
        o[0] = clamp8((int)(yy + 1.40200f * v + 0.5f));
        o[1] = clamp8((int)(yy - 0.34414f * u - 0.71414f * v + 0.5f));
        o[2] = clamp8((int)(yy + 1.77200f * u + 0.5f));
      }
    return rgb;
  }
};

}  // namespace jpeg
}  // namespace rth
