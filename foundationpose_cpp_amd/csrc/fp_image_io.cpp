// fp_image_io.cpp -- frame / dataset I/O of the acceptance harness without OpenCV.
//
// Replaces the helpers the reference's simple_tests use around Register / Track
// (simple_tests/include/tests/help_func.hpp: ReadRgbDepthMask :10-36, ReadRgbDepth :38-53, ReadCamK :108-129,
//  draw3DBoundingBox :55-106) for the dataset layout of test_data/download.md:6-15
//      <dir>/cam_K.txt  rgb/<id>.png  depth/<id>.png (u16, millimetres)  masks/<id>.png  mesh/
//   * rgb   : cv::imread + BGR2RGB                    -> u8 [H,W,3] RGB
//   * depth : IMREAD_UNCHANGED, convertTo f32, / 1000 -> f32 [H,W] metres
//   * mask  : IMREAD_UNCHANGED; 3-channel masks keep the first channel after BGR2RGB, i.e. the file's R channel
//   * cam_K : nine whitespace-separated numbers, row-major
// PNG: every bit depth and colour type, Adam7 interlacing (zlib inflate); mesh textures also BMP / PNM / TGA; writer: 8-bit RGB PNG.
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/foundationpose_amd.h"
#include "fp_internal.h"

namespace fp {

static uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// one PNG (sub-)image: undo the per-scanline filters of `h` rows of `rowbytes` bytes (each preceded by its filter byte)
static bool png_unfilter(const uint8_t *raw, size_t rowbytes, int h, int bpp, uint8_t *img) {
  for (int y = 0; y < h; y++) {
    const uint8_t *src = raw + (rowbytes + 1) * (size_t)y;
    const uint8_t ft = src[0];
    uint8_t *dst = img + rowbytes * (size_t)y;
    const uint8_t *up = y ? img + rowbytes * (size_t)(y - 1) : nullptr;
    for (size_t x = 0; x < rowbytes; x++) {
      int a = x >= (size_t)bpp ? dst[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= (size_t)bpp) ? up[x - bpp] : 0;
      int v = src[1 + x];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: {
          int pp = a + b - c, pa = std::abs(pp - a), pb = std::abs(pp - b), pc = std::abs(pp - c);
          v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          break;
        }
        default: return false;
      }
      dst[x] = (uint8_t)v;
    }
  }
  return true;
}

// Decoded samples, 16 bits each (8-bit files are widened without scaling; 1/2/4-bit grey is scaled to 0..255 like cv::imread,
// `bits` then reports 8), `ch` interleaved channels; palette expanded; Adam7-interlaced files are de-interlaced.
bool decode_png(const std::string &path, std::vector<uint16_t> &px, int &H, int &W, int &ch, int &bits) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (buf.size() < 33 || std::memcmp(buf.data(), sig, 8) != 0) return false;
  size_t pos = 8;
  int ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, plte;
  W = H = bits = 0;
  while (pos + 12 <= buf.size()) {
    uint32_t len = be32(&buf[pos]);
    if (pos + 12 + (size_t)len > buf.size()) return false;
    const uint8_t *d = &buf[pos + 8];
    if (!std::memcmp(&buf[pos + 4], "IHDR", 4) && len >= 13) {
      W = (int)be32(d); H = (int)be32(d + 4); bits = d[8]; ctype = d[9]; interlace = d[12];
    } else if (!std::memcmp(&buf[pos + 4], "PLTE", 4)) {
      plte.assign(d, d + len);
    } else if (!std::memcmp(&buf[pos + 4], "IDAT", 4)) {
      idat.insert(idat.end(), d, d + len);
    } else if (!std::memcmp(&buf[pos + 4], "IEND", 4)) {
      break;
    }
    pos += 12 + (size_t)len;
  }
  // IHDR is file-supplied: bound the image (16384^2 covers any camera frame or texture) before it sizes an allocation, and
  // keep every byte count inside zlib's 32-bit fields
  if (W <= 0 || H <= 0 || W > 16384 || H > 16384 || interlace > 1) return false;
  const int fch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if (!fch) return false;
  const bool sub = bits == 1 || bits == 2 || bits == 4;       // sub-byte samples: grey and palette images only
  if (!(bits == 8 || bits == 16 || (sub && (ctype == 0 || ctype == 3))) || (ctype == 3 && bits == 16)) return false;
  const int bpp = std::max(1, fch * bits / 8);                 // filter distance in bytes
  auto rowbytes = [&](int w) { return ((size_t)w * fch * bits + 7) / 8; };
  // passes: the whole image, or the seven Adam7 sub-images (x0, y0, dx, dy)
  static const int A7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
  struct Pass { int x0, y0, dx, dy, w, h; };
  std::vector<Pass> passes;
  if (!interlace) passes.push_back({0, 0, 1, 1, W, H});
  else
    for (auto &a : A7) {
      const int w = (W - a[0] + a[2] - 1) / a[2], h = (H - a[1] + a[3] - 1) / a[3];
      if (w > 0 && h > 0) passes.push_back({a[0], a[1], a[2], a[3], w, h});
    }
  size_t raw_size = 0;
  for (auto &ps : passes) raw_size += (rowbytes(ps.w) + 1) * (size_t)ps.h;
  if (raw_size > 0x7fffffffu || idat.size() > 0x7fffffffu) return false;
  std::vector<uint8_t> raw(raw_size);
  {
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (inflateInit(&zs) != Z_OK) return false;
    zs.next_in = idat.data();
    zs.avail_in = (uInt)idat.size();
    zs.next_out = raw.data();
    zs.avail_out = (uInt)raw.size();
    int rc = inflate(&zs, Z_FINISH);
    size_t got = zs.total_out;
    inflateEnd(&zs);
    if (!(rc == Z_STREAM_END || rc == Z_OK || rc == Z_BUF_ERROR) || got != raw.size()) return false;
  }
  // samples of the full image, one uint16 per sample (sub-byte samples unpacked, not yet scaled)
  std::vector<uint16_t> smp((size_t)W * H * fch);
  size_t off = 0;
  std::vector<uint8_t> img;
  for (auto &ps : passes) {
    const size_t rb = rowbytes(ps.w);
    img.assign(rb * (size_t)ps.h, 0);
    if (!png_unfilter(raw.data() + off, rb, ps.h, bpp, img.data())) return false;
    off += (rb + 1) * (size_t)ps.h;
    for (int y = 0; y < ps.h; y++) {
      const uint8_t *row = &img[rb * (size_t)y];
      uint16_t *dst = &smp[((size_t)(ps.y0 + y * ps.dy) * W + ps.x0) * fch];
      for (int x = 0; x < ps.w; x++, dst += (size_t)ps.dx * fch)
        for (int c = 0; c < fch; c++) {
          const size_t i = (size_t)x * fch + c;
          if (bits == 16) dst[c] = (uint16_t)((row[2 * i] << 8) | row[2 * i + 1]);   // big-endian samples
          else if (bits == 8) dst[c] = row[i];
          else dst[c] = (uint16_t)((row[i * bits / 8] >> (8 - bits - (int)((i * bits) % 8))) & ((1 << bits) - 1));
        }
    }
  }
  ch = ctype == 3 ? 3 : fch;
  const size_t n = (size_t)W * H;
  px.resize(n * ch);
  if (ctype == 3) {
    for (size_t i = 0; i < n; i++) {
      const size_t e = (size_t)smp[i] * 3;
      if (e + 2 >= plte.size()) return false;
      px[i * 3] = plte[e]; px[i * 3 + 1] = plte[e + 1]; px[i * 3 + 2] = plte[e + 2];
    }
    bits = 8;
  } else if (sub) {
    const int mx = (1 << bits) - 1;
    for (size_t i = 0; i < n; i++) px[i] = (uint16_t)(smp[i] * 255 / mx);
    bits = 8;
  } else {
    px = std::move(smp);
  }
  return true;
}

// ---- the other texture containers cv::imread reads without an external codec: BMP, binary / ASCII PNM, TGA ----
static bool decode_bmp(const std::vector<uint8_t> &b, std::vector<uint8_t> &rgb, int &H, int &W) {
  auto le16 = [&](size_t o) { return (uint32_t)b[o] | ((uint32_t)b[o + 1] << 8); };
  auto le32 = [&](size_t o) { return le16(o) | (le16(o + 2) << 16); };
  if (b.size() < 54 || b[0] != 'B' || b[1] != 'M') return false;
  const uint32_t data_off = le32(10), hdr = le32(14);
  if (hdr < 40) return false;
  const int w = (int)le32(18), hs = (int)le32(22);
  const int bpp = (int)le16(28);
  const uint32_t comp = le32(30);
  const bool bottom_up = hs > 0;
  const int h = std::abs(hs);
  if (w <= 0 || h <= 0 || w > 16384 || h > 16384 || !(comp == 0 || (comp == 3 && bpp == 32)) || !(bpp == 8 || bpp == 24 || bpp == 32)) return false;
  const size_t stride = (((size_t)w * bpp + 31) / 32) * 4;
  if ((size_t)data_off + stride * h > b.size()) return false;
  const size_t pal = 14 + (size_t)hdr;
  uint32_t ncol = le32(46);
  if (bpp == 8 && ncol == 0) ncol = 256;
  if (bpp == 8 && pal + (size_t)ncol * 4 > b.size()) return false;
  W = w; H = h;
  rgb.resize((size_t)w * h * 3);
  for (int y = 0; y < h; y++) {
    const uint8_t *row = &b[data_off + stride * (size_t)(bottom_up ? h - 1 - y : y)];
    uint8_t *dst = &rgb[(size_t)y * w * 3];
    for (int x = 0; x < w; x++) {
      const uint8_t *q = bpp == 8 ? &b[pal + (size_t)std::min<uint32_t>(row[x], ncol - 1) * 4] : row + (size_t)x * (bpp / 8);   // B, G, R(, A)
      dst[x * 3] = q[2]; dst[x * 3 + 1] = q[1]; dst[x * 3 + 2] = q[0];
    }
  }
  return true;
}

static bool decode_pnm(const std::vector<uint8_t> &b, std::vector<uint8_t> &rgb, int &H, int &W) {
  if (b.size() < 7 || b[0] != 'P' || b[1] < '2' || b[1] > '6' || b[1] == '4') return false;
  const int kind = b[1] - '0';          // 2 / 5 grey, 3 / 6 colour; 2, 3 ASCII
  size_t pos = 2;
  auto next_int = [&](long &v) {
    for (;;) {
      while (pos < b.size() && std::isspace(b[pos])) pos++;
      if (pos < b.size() && b[pos] == '#') { while (pos < b.size() && b[pos] != '\n') pos++; continue; }
      break;
    }
    if (pos >= b.size() || !std::isdigit(b[pos])) return false;
    v = 0;
    while (pos < b.size() && std::isdigit(b[pos]) && v < (1L << 40)) v = v * 10 + (b[pos++] - '0');
    return true;
  };
  long w, h, mx;
  if (!next_int(w) || !next_int(h) || !next_int(mx) || w <= 0 || h <= 0 || w > 16384 || h > 16384 || mx <= 0 || mx > 65535) return false;
  const int ch = (kind == 3 || kind == 6) ? 3 : 1;
  const size_t n = (size_t)w * h * ch;
  std::vector<long> v(n);
  if (kind >= 5) {
    pos++;                               // the single whitespace byte after maxval
    const int bs = mx > 255 ? 2 : 1;
    if (pos + n * bs > b.size()) return false;
    for (size_t i = 0; i < n; i++) v[i] = bs == 2 ? ((long)b[pos + 2 * i] << 8 | b[pos + 2 * i + 1]) : b[pos + i];
  } else {
    for (size_t i = 0; i < n; i++) if (!next_int(v[i])) return false;
  }
  W = (int)w; H = (int)h;
  rgb.resize((size_t)w * h * 3);
  for (size_t i = 0; i < (size_t)w * h; i++)
    for (int c = 0; c < 3; c++) {
      const long s = std::min(v[i * ch + (ch == 3 ? c : 0)], mx);
      rgb[i * 3 + c] = (uint8_t)(mx == 255 ? s : mx > 255 ? (s >> 8) : s * 255 / mx);   // 16-bit -> 8-bit like cv::imread (>> 8)
    }
  return true;
}

static bool decode_tga(const std::vector<uint8_t> &b, std::vector<uint8_t> &rgb, int &H, int &W) {
  if (b.size() < 18) return false;
  const int idlen = b[0], cmap = b[1], type = b[2], w = b[12] | (b[13] << 8), h = b[14] | (b[15] << 8), bpp = b[16], desc = b[17];
  const bool rle = type == 10 || type == 11, grey = type == 3 || type == 11;
  if (cmap != 0 || !(type == 2 || type == 3 || type == 10 || type == 11) || w <= 0 || h <= 0) return false;
  if (!((grey && bpp == 8) || (!grey && (bpp == 24 || bpp == 32)))) return false;
  const int bs = bpp / 8;
  size_t pos = 18 + (size_t)idlen;
  std::vector<uint8_t> pix((size_t)w * h * bs);
  if (!rle) {
    if (pos + pix.size() > b.size()) return false;
    std::memcpy(pix.data(), &b[pos], pix.size());
  } else {
    size_t o = 0;
    while (o < pix.size()) {
      if (pos >= b.size()) return false;
      const int hd = b[pos++], cnt = (hd & 127) + 1;
      if (hd & 128) {
        if (pos + bs > b.size() || o + (size_t)cnt * bs > pix.size()) return false;
        for (int k = 0; k < cnt; k++, o += bs) std::memcpy(&pix[o], &b[pos], bs);
        pos += bs;
      } else {
        if (pos + (size_t)cnt * bs > b.size() || o + (size_t)cnt * bs > pix.size()) return false;
        std::memcpy(&pix[o], &b[pos], (size_t)cnt * bs);
        pos += (size_t)cnt * bs; o += (size_t)cnt * bs;
      }
    }
  }
  const bool top_down = (desc & 0x20) != 0, right_left = (desc & 0x10) != 0;
  W = w; H = h;
  rgb.resize((size_t)w * h * 3);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      const uint8_t *q = &pix[((size_t)(top_down ? y : h - 1 - y) * w + (right_left ? w - 1 - x : x)) * bs];
      uint8_t *d = &rgb[((size_t)y * w + x) * 3];
      if (grey) d[0] = d[1] = d[2] = q[0];
      else { d[0] = q[2]; d[1] = q[1]; d[2] = q[0]; }
    }
  return true;
}

// Texture file -> 8-bit RGB the way cv::imread(path) + BGR2RGB delivers it (assimp_mesh_loader.cpp:216-223): grey replicated, alpha
// dropped, palette expanded, 16-bit samples >> 8.  Containers: PNG (every bit depth / colour type, Adam7), BMP, PNM, TGA.  JPEG and
// the rest of cv::imread's list need a codec this library does not carry: *why says so.
bool load_texture_rgb(const std::string &path, std::vector<uint8_t> &rgb, int &H, int &W, std::string *why) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { if (why) *why = "cannot open"; return false; }
  std::vector<uint8_t> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  f.close();
  auto fail = [&](const char *m) { if (why) *why = m; return false; };
  if (b.size() >= 8 && b[0] == 0x89 && b[1] == 'P' && b[2] == 'N' && b[3] == 'G') {
    std::vector<uint16_t> px;
    int ch = 0, bits = 0;
    if (!decode_png(path, px, H, W, ch, bits)) return fail("corrupt or unsupported PNG");
    const int sh = bits == 16 ? 8 : 0;
    rgb.resize((size_t)W * H * 3);
    for (size_t i = 0; i < (size_t)W * H; i++) {
      const uint16_t *p = &px[i * ch];
      const bool grey = ch <= 2;
      rgb[i * 3] = (uint8_t)(p[0] >> sh);
      rgb[i * 3 + 1] = (uint8_t)((grey ? p[0] : p[1]) >> sh);
      rgb[i * 3 + 2] = (uint8_t)((grey ? p[0] : p[2]) >> sh);
    }
    return true;
  }
  if (b.size() >= 3 && b[0] == 0xFF && b[1] == 0xD8 && b[2] == 0xFF) return fail("JPEG textures are not supported (no JPEG codec in this library): convert the texture to PNG");
  if (b.size() >= 2 && b[0] == 'B' && b[1] == 'M') return decode_bmp(b, rgb, H, W) || fail("corrupt or unsupported BMP (supported: uncompressed 8 / 24 / 32 bit)");
  if (b.size() >= 2 && b[0] == 'P' && b[1] >= '1' && b[1] <= '6') return decode_pnm(b, rgb, H, W) || fail("corrupt or unsupported PNM (supported: P2 / P3 / P5 / P6)");
  const std::string ext = path.size() >= 4 ? path.substr(path.size() - 4) : "";
  if (ext == ".tga" || ext == ".TGA") return decode_tga(b, rgb, H, W) || fail("corrupt or unsupported TGA (supported: true-colour / grey, raw or RLE)");
  return fail("unknown image container (supported: PNG, BMP, PNM, TGA)");
}

// 8-bit view as RGB (grey replicated, alpha dropped); kept for callers that insist on an 8-bit PNG
bool load_png_rgb(const std::string &path, std::vector<uint8_t> &rgb, int &H, int &W) {
  std::vector<uint16_t> px;
  int ch = 0, bits = 0;
  if (!decode_png(path, px, H, W, ch, bits) || bits != 8) return false;
  rgb.resize((size_t)W * H * 3);
  for (size_t i = 0; i < (size_t)W * H; i++) {
    const uint16_t *p = &px[i * ch];
    const bool grey = ch <= 2;
    rgb[i * 3] = (uint8_t)p[0];
    rgb[i * 3 + 1] = (uint8_t)(grey ? p[0] : p[1]);
    rgb[i * 3 + 2] = (uint8_t)(grey ? p[0] : p[2]);
  }
  return true;
}

static void put_be32(std::vector<uint8_t> &o, uint32_t v) {
  o.push_back(v >> 24); o.push_back((v >> 16) & 255); o.push_back((v >> 8) & 255); o.push_back(v & 255);
}

static void png_chunk(std::vector<uint8_t> &o, const char *type, const std::vector<uint8_t> &d) {
  put_be32(o, (uint32_t)d.size());
  size_t s = o.size();
  o.insert(o.end(), type, type + 4);
  o.insert(o.end(), d.begin(), d.end());
  put_be32(o, (uint32_t)crc32(0, &o[s], (uInt)(o.size() - s)));
}

}  // namespace fp

extern "C" {

static int fp_image_read_png_impl(const char *path, int *H, int *W, int *channels, int *bit_depth, uint16_t *out, size_t out_capacity) {
  std::vector<uint16_t> px;
  int h, w, ch, bits;
  FP_CHECK(path && fp::decode_png(path, px, h, w, ch, bits), std::string("Failed reading png from path : ") + (path ? path : "(null)"));
  if (H) *H = h;
  if (W) *W = w;
  if (channels) *channels = ch;
  if (bit_depth) *bit_depth = bits;
  if (out) {
    FP_CHECK(out_capacity >= px.size(), "fp_image_read_png: output buffer too small");
    std::memcpy(out, px.data(), px.size() * sizeof(uint16_t));
  }
  return 0;
}
int fp_image_read_png(const char *path, int *H, int *W, int *channels, int *bit_depth, uint16_t *out, size_t out_capacity) {
  try {
    return fp_image_read_png_impl(path, H, W, channels, bit_depth, out, out_capacity);
  } catch (const std::exception &e) {  // nothing may unwind through the C ABI
    fp::set_error(std::string("fp_image_read_png: ") + e.what());
    return 1;
  }
}


static int read_frame_part(const char *path, const char *what, std::vector<uint16_t> &px, int &h, int &w, int &ch, int &bits) {
  FP_CHECK(path && fp::decode_png(path, px, h, w, ch, bits), std::string("Failed reading ") + what + " from path : " + (path ? path : "(null)"));
  return 0;
}

static int fp_frame_size_impl(const char *rgb_path, int *H, int *W) {
  std::vector<uint16_t> px;
  int h, w, ch, bits;
  if (read_frame_part(rgb_path, "rgb", px, h, w, ch, bits)) return 1;
  *H = h;
  *W = w;
  return 0;
}
int fp_frame_size(const char *rgb_path, int *H, int *W) {
  try {
    return fp_frame_size_impl(rgb_path, H, W);
  } catch (const std::exception &e) {  // nothing may unwind through the C ABI
    fp::set_error(std::string("fp_frame_size: ") + e.what());
    return 1;
  }
}


static int fp_read_rgb_depth_mask_impl(const char *rgb_path, const char *depth_path, const char *mask_path, int H, int W,
                           uint8_t *rgb, float *depth, uint8_t *mask) {
  std::vector<uint16_t> px;
  int h, w, ch, bits;
  const size_t n = (size_t)H * W;
  if (rgb) {
    if (read_frame_part(rgb_path, "rgb", px, h, w, ch, bits)) return 1;
    FP_CHECK(h == H && w == W && bits == 8, std::string("rgb image has an unexpected size or bit depth: ") + rgb_path);
    for (size_t i = 0; i < n; i++)
      for (int c = 0; c < 3; c++) rgb[i * 3 + c] = (uint8_t)px[i * ch + (ch >= 3 ? c : 0)];
  }
  if (depth) {
    if (read_frame_part(depth_path, "depth", px, h, w, ch, bits)) return 1;
    FP_CHECK(h == H && w == W, std::string("depth image has an unexpected size: ") + depth_path);
    for (size_t i = 0; i < n; i++) depth[i] = (float)px[i * ch] / 1000.f;  // convertTo(CV_32FC1) then / 1000.f
  }
  if (mask) {
    if (read_frame_part(mask_path, "mask", px, h, w, ch, bits)) return 1;
    FP_CHECK(h == H && w == W, std::string("mask image has an unexpected size: ") + mask_path);
    for (size_t i = 0; i < n; i++) {
      uint16_t v = px[i * ch];  // single channel, or the file's R channel
      mask[i] = (uint8_t)(bits == 16 ? (v > 255 ? 255 : v) : v);
    }
  }
  return 0;
}
int fp_read_rgb_depth_mask(const char *rgb_path, const char *depth_path, const char *mask_path, int H, int W,
                           uint8_t *rgb, float *depth, uint8_t *mask) {
  try {
    return fp_read_rgb_depth_mask_impl(rgb_path, depth_path, mask_path, H, W, rgb, depth, mask);
  } catch (const std::exception &e) {  // nothing may unwind through the C ABI
    fp::set_error(std::string("fp_read_rgb_depth_mask: ") + e.what());
    return 1;
  }
}


static int fp_read_cam_k_impl(const char *cam_K_path, float K[9]) {
  std::ifstream f(cam_K_path ? cam_K_path : "");
  FP_CHECK((bool)f, std::string("Failed open file : ") + (cam_K_path ? cam_K_path : "(null)"));
  for (int i = 0; i < 9; i++) {
    double v;
    FP_CHECK((bool)(f >> v), std::string("cam_K file holds fewer than 9 numbers: ") + cam_K_path);
    K[i] = (float)v;
  }
  return 0;
}
int fp_read_cam_k(const char *cam_K_path, float K[9]) {
  try {
    return fp_read_cam_k_impl(cam_K_path, K);
  } catch (const std::exception &e) {  // nothing may unwind through the C ABI
    fp::set_error(std::string("fp_read_cam_k: ") + e.what());
    return 1;
  }
}


static int fp_image_write_png_rgb_impl(const char *path, const uint8_t *rgb, int H, int W) {
  FP_CHECK(path && rgb && H > 0 && W > 0, "fp_image_write_png_rgb: bad arguments");
  std::vector<uint8_t> raw((size_t)H * (W * 3 + 1));
  for (int y = 0; y < H; y++) {
    raw[(size_t)y * (W * 3 + 1)] = 0;
    std::memcpy(&raw[(size_t)y * (W * 3 + 1) + 1], rgb + (size_t)y * W * 3, (size_t)W * 3);
  }
  uLongf clen = compressBound((uLong)raw.size());
  std::vector<uint8_t> comp(clen);
  FP_CHECK(compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) == Z_OK, "png deflate failed");
  comp.resize(clen);
  std::vector<uint8_t> o = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  std::vector<uint8_t> ihdr;
  fp::put_be32(ihdr, (uint32_t)W);
  fp::put_be32(ihdr, (uint32_t)H);
  ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});
  fp::png_chunk(o, "IHDR", ihdr);
  fp::png_chunk(o, "IDAT", comp);
  fp::png_chunk(o, "IEND", {});
  FILE *fo = std::fopen(path, "wb");
  FP_CHECK(fo != nullptr, std::string("cannot write ") + path);
  size_t wr = std::fwrite(o.data(), 1, o.size(), fo);
  std::fclose(fo);
  FP_CHECK(wr == o.size(), std::string("short write to ") + path);
  return 0;
}
int fp_image_write_png_rgb(const char *path, const uint8_t *rgb, int H, int W) {
  try {
    return fp_image_write_png_rgb_impl(path, rgb, H, W);
  } catch (const std::exception &e) {  // nothing may unwind through the C ABI
    fp::set_error(std::string("fp_image_write_png_rgb: ") + e.what());
    return 1;
  }
}


// draw3DBoundingBox (help_func.hpp:55-106): the 8 corners (+-dimension/2) through `pose` (column-major bbox->camera, i.e.
// ConvertPoseMesh2BBox(pose, loader)), pinhole projection with fx, fy, cx, cy, 12 green edges of thickness 2.
int fp_draw_bbox3d(uint8_t *rgb, int H, int W, const float K[9], const float pose[16], const float dimension[3]) {
  FP_CHECK(rgb && K && pose && dimension, "fp_draw_bbox3d: null argument");
  const float l = dimension[0] / 2, w = dimension[1] / 2, h = dimension[2] / 2;
  const float pts[8][3] = {{-l, -w, h}, {l, -w, h}, {l, w, h}, {-l, w, h}, {-l, -w, -h}, {l, -w, -h}, {l, w, -h}, {-l, w, -h}};
  float u[8], v[8];
  for (int i = 0; i < 8; i++) {
    float c[3];
    for (int r = 0; r < 3; r++) c[r] = pose[r] * pts[i][0] + pose[4 + r] * pts[i][1] + pose[8 + r] * pts[i][2] + pose[12 + r];
    u[i] = K[0] * (c[0] / c[2]) + K[2];
    v[i] = K[4] * (c[1] / c[2]) + K[5];
  }
  static const int edges[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
  for (auto &e : edges) {
    float x0 = u[e[0]], y0 = v[e[0]], x1 = u[e[1]], y1 = v[e[1]];
    if (!std::isfinite(x0 + y0 + x1 + y1)) continue;
    int steps = (int)std::ceil(std::max(std::fabs(x1 - x0), std::fabs(y1 - y0)));
    steps = std::min(std::max(steps, 1), 4 * (H + W));
    for (int s = 0; s <= steps; s++) {
      float t = (float)s / steps;
      int cx = (int)std::lround(x0 + t * (x1 - x0)), cy = (int)std::lround(y0 + t * (y1 - y0));
      for (int dy = 0; dy < 2; dy++)
        for (int dx = 0; dx < 2; dx++) {
          int px = cx + dx, py = cy + dy;
          if (px < 0 || py < 0 || px >= W || py >= H) continue;
          uint8_t *p = rgb + ((size_t)py * W + px) * 3;
          p[0] = 0; p[1] = 255; p[2] = 0;
        }
    }
  }
  return 0;
}

}  // extern "C"
