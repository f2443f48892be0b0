// Image output for the render buffer: RGB8, row 0 = bottom of the image.
// reference: writeImage dispatch on the file extension (src/fileio/images.cc:59-68), PNG rows are
// written top-down so the buffer is flipped (src/fileio/pngimage.cpp:264-265), BMP is stored
// bottom-up as it is (src/fileio/bitmap.cpp:95-149).
#pragma once
#include <cstdint>
#include <string>

namespace cpt {

// Returns false and fills err on failure.  The extension (.png or .bmp, any case) picks the format.
bool write_image(const std::string& path, int width, int height, const uint8_t* rgb_bottom_up, std::string& err);

}  // namespace cpt
