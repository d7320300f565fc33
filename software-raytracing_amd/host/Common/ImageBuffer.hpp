// ImageBuffer.hpp - owning row-major 2-D buffer with the reference's interface (Sources/Common/ImageBuffer.hpp:13-220):
// top-left origin, operator[](row), setPoint with optional bounds check, getData/getWidth/getHeight/getSize(bytes)/clear.
// The GPU path writes the whole frame through getData() (4 bytes per RGBQUAD, B,G,R,reserved).
#pragma once
#include <algorithm>
#include <cstring>
#include <vector>

template <typename T> class ImageBuffer {
    unsigned width_ = 0, height_ = 0;
    std::vector<T> data_;
public:
    ImageBuffer() = default;
    ImageBuffer(unsigned width, unsigned height, const T& clear) : width_(width), height_(height), data_(static_cast<size_t>(width) * height, clear) {}
    T* operator[](int y) { return data_.data() + static_cast<size_t>(width_) * y; }
    unsigned getSize() const { return static_cast<unsigned>(data_.size() * sizeof(T)); }
    void clear(const T& v) { std::fill(data_.begin(), data_.end(), v); }
    T* getData() { return data_.empty() ? nullptr : data_.data(); }
    unsigned getWidth() const { return width_; }
    unsigned getHeight() const { return height_; }
    bool isPointIn(int x, int y) const { return !data_.empty() && x >= 0 && y >= 0 && static_cast<unsigned>(x) < width_ && static_cast<unsigned>(y) < height_; }
    void setPoint(unsigned x, unsigned y, T color, bool safe = true) { if (safe && !isPointIn(static_cast<int>(x), static_cast<int>(y))) return; (*this)[static_cast<int>(y)][x] = color; }
};
