// scanline.h -- drop-in for the reference's src/scanline.h: absolute and relative (R, t) of one image row.
#ifndef RSDSFM_HOST_SCANLINE_H
#define RSDSFM_HOST_SCANLINE_H

#include "rsdsfm_eigen_lite.hpp"

class Scanline {
public:
    Scanline() : rotation_(rsdsfm::lite::Matrix3d::Zero()), translation_(rsdsfm::lite::Vector3d::Zero()) {}
    Scanline(const rsdsfm::lite::Matrix3d& rotation, const rsdsfm::lite::Vector3d& translation) : rotation_(rotation), translation_(translation) {}
    const rsdsfm::lite::Matrix3d& getRotation() const { return rotation_; }
    const rsdsfm::lite::Matrix3d& getRelativeRotation() const { return relative_rotation_; }
    const rsdsfm::lite::Vector3d& getTranslation() const { return translation_; }
    const rsdsfm::lite::Vector3d& getRelativeTranslation() const { return relative_translation_; }
    void setRotation(const rsdsfm::lite::Matrix3d& rotation) { rotation_ = rotation; }
    void setTranslation(const rsdsfm::lite::Vector3d& translation) { translation_ = translation; }
    void setRelativeRotation(const rsdsfm::lite::Matrix3d& rotation) { relative_rotation_ = rotation; }
    void setRelativeTranslation(const rsdsfm::lite::Vector3d& translation) { relative_translation_ = translation; }

private:
    rsdsfm::lite::Matrix3d rotation_;
    rsdsfm::lite::Vector3d translation_;
    rsdsfm::lite::Matrix3d relative_rotation_;
    rsdsfm::lite::Vector3d relative_translation_;
};

#endif
